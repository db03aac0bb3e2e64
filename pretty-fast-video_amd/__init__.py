"""pretty-fast-video_amd -- MI355X (gfx950) transform/motion hot path of Pretty Fast Video.

Host-side mirror (Python, over the C ABI of libpfv_hip.so) of the reference's operator
interface for this path: ``VideoPlane`` / ``VideoFrame`` (src/plane.rs, src/frame.rs), the
plane-level operators of ``impl VideoPlane`` (src/common.rs:351-521) and the hot-path half
of ``Encoder`` / ``Decoder`` (src/enc.rs, src/dec.rs).  All arithmetic runs in the HIP
kernels under ``csrc/``; this package only marshals buffers.

The directory name contains a hyphen, so import it through
``__graft_entry__.load_package()`` (alias ``pretty_fast_video_amd``).
"""
from . import _lib
from ._lib import PfvError
from .context import Context, Graph
from .plane import VideoPlane, EncodedIPlane, EncodedPPlane
from .frame import VideoFrame
from .session import EncoderSession, DecoderSession, qtables_from_quality
from .enc import BatchEncoder, Encoder, GopEncoder
from .dec import BatchDecoder, Decoder, DecodeError, GopDecoder
from .synth import SyntheticStream

__all__ = ["Context", "Graph", "VideoPlane", "VideoFrame", "EncodedIPlane", "EncodedPPlane", "EncoderSession",
           "DecoderSession", "Encoder", "BatchEncoder", "GopEncoder", "Decoder", "BatchDecoder", "GopDecoder", "DecodeError", "qtables_from_quality", "PfvError", "SyntheticStream", "_lib"]
