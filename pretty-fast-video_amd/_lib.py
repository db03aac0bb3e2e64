"""ctypes loader for libpfv_hip.so (the C ABI declared in include/pfv_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc --offload-arch=gfx950).
There is no fallback and no redirection: the in-tree ``libpfv_hip.so`` is what loads; if it is
missing or no GPU is visible, the calls fail loudly.  (The non-GPU test suite and the A/B scripts
swap other builds of the same C ABI in from the outside: tests/libswitch.py.)
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_size_t, c_uint8, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libpfv_hip.so")

PFV_OK = 0
PFV_ERR_BAD_ARG = -1
PFV_ERR_HIP = -2
PFV_ERR_NOMEM = -3
PFV_ERR_BAD_MV = -4
PFV_ERR_NO_DEVICE = -5
PFV_ERR_FORMAT = -6
PFV_ERR_VERSION = -7
PFV_ERR_IO = -8
PFV_ERR_STATE = -9

# pfv_ctx_set_option
PFV_COMM_SUM, PFV_COMM_MAX = 0, 1
PFV_OPT_ENC_TRANSFORM = 1
PFV_OPT_TILE_COMPACTION = 2
PFV_OPT_LANE_MAPPING = 3
PFV_LANES_AUTO, PFV_LANES_PER_MB_8, PFV_LANES_PER_MB_16 = 0, 1, 2
PFV_OPT_ENTROPY_DECODE = 4
PFV_OPT_ENTDEC_LANE_BITS, PFV_OPT_ENTDEC_LAUNCHES, PFV_OPT_ENTDEC_INNER_ROUNDS = 5, 6, 7
PFV_ENTROPY_DECODE_AUTO, PFV_ENTROPY_DECODE_HOST, PFV_ENTROPY_DECODE_DEVICE = 0, 1, 2
PFV_ENC_TRANSFORM_AUTO, PFV_ENC_TRANSFORM_INT = 0, 1


class PfvError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"pfv_hip error {code}: {msg}")
        self.code = code


# (name, restype, argtypes) -- must list every PFV_API symbol of include/pfv_hip.h
_P = c_void_p
SIGNATURES = [
    ("pfv_device_count", c_int, []),
    ("pfv_ctx_create", c_int, [c_int, POINTER(_P)]),
    ("pfv_ctx_create_prio", c_int, [c_int, c_int, POINTER(_P)]),
    ("pfv_ctx_destroy", None, [_P]),
    ("pfv_ctx_pci_bus_id", c_int, [_P, _P, c_int]),
    ("pfv_ctx_sync", c_int, [_P]),
    ("pfv_device_sync", c_int, [_P]),
    ("pfv_ctx_stream", _P, [_P]),
    ("pfv_last_error", c_char_p, [_P]),
    ("pfv_version", c_char_p, []),
    ("pfv_ctx_set_option", c_int, [_P, c_int, c_int]),
    ("pfv_ctx_get_option", c_int, [_P, c_int, POINTER(c_int)]),
    ("pfv_event_create", c_int, [_P, POINTER(_P)]),
    ("pfv_event_record", c_int, [_P]),
    ("pfv_event_elapsed_ms", c_int, [_P, _P, POINTER(c_float)]),
    ("pfv_event_destroy", None, [_P]),
    ("pfv_ctx_wait_event", c_int, [_P, _P]),
    ("pfv_graph_begin", c_int, [_P]),
    ("pfv_graph_end", c_int, [_P, POINTER(_P)]),
    ("pfv_graph_launch", c_int, [_P]),
    ("pfv_graph_destroy", None, [_P]),
    ("pfv_pad16", c_int, [c_int]),
    ("pfv_qtables_from_quality", c_int, [c_int, _P, _P, _P, _P, POINTER(c_float)]),
    ("pfv_encode_plane", c_int, [_P, _P, c_int, c_int, _P, c_uint8, _P]),
    ("pfv_encode_plane_delta", c_int, [_P, _P, c_int, c_int, _P, _P, c_float, c_uint8, _P, _P, _P]),
    ("pfv_decode_plane_into", c_int, [_P, _P, c_int, c_int, _P, _P]),
    ("pfv_decode_plane_delta", c_int, [_P, _P, _P, _P, c_int, c_int, _P, _P, _P]),
    ("pfv_decode_plane_delta_into", c_int, [_P, _P, _P, _P, c_int, c_int, _P, _P]),
    ("pfv_blit_dev", c_int, [_P, _P, c_int, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    ("pfv_reduce_dev", c_int, [_P, _P, _P, c_int, c_int]),
    ("pfv_double_dev", c_int, [_P, _P, _P, c_int, c_int]),
    ("pfv_rgb_to_yuv420_dev", c_int, [_P, _P, c_int, c_int, _P]),
    ("pfv_yuv420_to_rgb_dev", c_int, [_P, _P, c_int, c_int, _P]),
    ("pfv_comm_unique_id", c_int, [_P]),
    ("pfv_comm_init", c_int, [_P, c_int, c_int, _P, POINTER(_P)]),
    ("pfv_comm_rank", c_int, [_P]),
    ("pfv_comm_world", c_int, [_P]),
    ("pfv_comm_broadcast_dev", c_int, [_P, _P, c_size_t, c_int]),
    ("pfv_comm_allreduce_f64_dev", c_int, [_P, _P, c_size_t, c_int]),
    ("pfv_comm_allgather_dev", c_int, [_P, _P, _P, c_size_t]),
    ("pfv_comm_allreduce_f64", c_int, [_P, _P, c_size_t, c_int]),
    ("pfv_comm_barrier", c_int, [_P]),
    ("pfv_comm_destroy", None, [_P]),
    ("pfv_synth_frames_dev", c_int, [_P, c_int, c_int, c_int, _P, c_int, _P]),
    ("pfv_synth_frames_kind_dev", c_int, [_P, c_int, c_int, c_int, _P, c_int, c_int, _P]),
    ("pfv_dev_alloc", c_int, [_P, c_size_t, POINTER(_P)]),
    ("pfv_dev_copy", c_int, [_P, _P, _P, c_size_t]),
    ("pfv_dev_free", c_int, [_P, _P]),
    ("pfv_host_alloc", c_int, [_P, c_size_t, POINTER(_P)]),
    ("pfv_host_free", c_int, [_P, _P]),
    ("pfv_dev_upload", c_int, [_P, _P, _P, c_size_t]),
    ("pfv_dev_download", c_int, [_P, _P, _P, c_size_t]),
    ("pfv_frame_bytes", c_size_t, [c_int, c_int]),
    ("pfv_padded_frame_bytes", c_size_t, [c_int, c_int]),
    ("pfv_total_blocks", c_int, [c_int, c_int]),
    ("pfv_enc_session_create", c_int, [_P, c_int, c_int, c_int, c_int, POINTER(_P)]),
    ("pfv_enc_session_destroy", None, [_P]),
    ("pfv_enc_iframe_dev", c_int, [_P, _P, _P]),
    ("pfv_enc_pframe_dev", c_int, [_P, _P, _P, _P, _P]),
    ("pfv_enc_iframe", c_int, [_P, _P, _P]),
    ("pfv_enc_pframe", c_int, [_P, _P, _P, _P, _P]),
    ("pfv_enc_session_set_frame_stride", c_int, [_P, c_size_t]),
    ("pfv_enc_session_set_window", c_int, [_P, c_int, c_int]),
    ("pfv_enc_prev_frame_dev", _P, [_P, c_int]),
    ("pfv_enc_prev_frame", c_int, [_P, _P]),
    ("pfv_payload_worst_case", c_size_t, [c_int, c_int]),
    ("pfv_enc_entropy_enable", c_int, [_P, c_size_t]),
    ("pfv_enc_entropy_set_async", c_int, [_P, c_int]),
    ("pfv_enc_entropy_join", c_int, [_P]),
    ("pfv_enc_pack_iframe_dev", c_int, [_P, _P]),
    ("pfv_enc_pack_pframe_dev", c_int, [_P, _P, _P, _P]),
    ("pfv_enc_payload_sizes", c_int, [_P, _P]),
    ("pfv_enc_payloads_fetch", c_int, [_P, _P, c_size_t, _P, _P]),
    ("pfv_enc_payload_dev", _P, [_P, c_int]),
    ("pfv_enc_payload_capacity", c_size_t, [_P]),
    ("pfv_enc_payload_fetch", c_int, [_P, c_int, _P, c_size_t]),
    ("pfv_dec_iframe_sparse", c_int, [_P, _P, _P, c_size_t, _P]),
    ("pfv_dec_pframe_sparse", c_int, [_P, _P, _P, _P, _P, c_size_t, _P]),
    ("pfv_dec_iframe_lists_dev", c_int, [_P, _P, _P, _P]),
    ("pfv_dec_pframe_lists_dev", c_int, [_P, _P, _P, _P, _P, _P]),
    ("pfv_coef_lists_from_dense", c_int, [_P, _P, c_int, _P, c_size_t, _P, POINTER(c_size_t)]),
    ("pfv_dec_session_create", c_int, [_P, c_int, c_int, _P, c_int, c_int, POINTER(_P)]),
    ("pfv_dec_session_destroy", None, [_P]),
    ("pfv_dec_iframe_dev", c_int, [_P, _P, _P]),
    ("pfv_dec_pframe_dev", c_int, [_P, _P, _P, _P, _P]),
    ("pfv_dec_iframe", c_int, [_P, _P, _P]),
    ("pfv_dec_pframe", c_int, [_P, _P, _P, _P, _P]),
    ("pfv_dec_get_frame_dev", c_int, [_P, _P]),
    ("pfv_dec_set_output_dev", c_int, [_P, _P]),
    ("pfv_dec_set_output_strided_dev", c_int, [_P, _P, c_size_t]),
    ("pfv_dec_session_set_window", c_int, [_P, c_int, c_int]),
    ("pfv_dec_get_frame", c_int, [_P, _P]),
    ("pfv_dec_framebuffer", c_int, [_P, _P]),
    ("pfv_dec_check", c_int, [_P]),
    ("pfv_encoder_create", c_int, [_P, c_int, c_int, c_int, c_int, POINTER(_P)]),
    ("pfv_encoder_encode_iframe", c_int, [_P, _P, _P, _P]),
    ("pfv_encoder_encode_pframe", c_int, [_P, _P, _P, _P]),
    ("pfv_encoder_encode_dropframe", c_int, [_P]),
    ("pfv_encoder_finish", c_int, [_P]),
    ("pfv_encoder_bytes", c_int, [_P, POINTER(_P), POINTER(c_size_t)]),
    ("pfv_encoder_drain", c_int, [_P, POINTER(_P), POINTER(c_size_t)]),
    ("pfv_encoder_destroy", None, [_P]),
    ("pfv_encoder_set_device_entropy", c_int, [_P, c_int]),
    ("pfv_batch_encoder_create", c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P, POINTER(_P)]),
    ("pfv_batch_encoder_frames", _P, [_P]),
    ("pfv_batch_encoder_encode", c_int, [_P, c_int, _P]),
    ("pfv_batch_encoder_flush", c_int, [_P]),
    ("pfv_batch_encoder_finish", c_int, [_P]),
    ("pfv_batch_encoder_take", c_int, [_P, c_int, POINTER(_P), POINTER(c_size_t)]),
    ("pfv_batch_encoder_destroy", None, [_P]),
    ("pfv_batch_decoder_create", c_int, [_P, _P, _P, c_int, c_int, POINTER(_P)]),
    ("pfv_batch_decoder_width", c_int, [_P]),
    ("pfv_batch_decoder_height", c_int, [_P]),
    ("pfv_batch_decoder_framerate", c_int, [_P]),
    ("pfv_batch_decoder_dense_steps", ctypes.c_long, [_P]),
    ("pfv_batch_decoder_advance", c_int, [_P, POINTER(_P)]),
    ("pfv_batch_decoder_destroy", None, [_P]),
    ("pfv_gop_encoder_create", c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, c_size_t, POINTER(_P)]),
    ("pfv_gop_encoder_encode_iframe", c_int, [_P, _P, _P, _P]),
    ("pfv_gop_encoder_encode_pframe", c_int, [_P, _P, _P, _P]),
    ("pfv_gop_encoder_set_frames_by_reference", c_int, [_P, c_int]),
    ("pfv_gop_encoder_encode_iframe_dev", c_int, [_P, _P]),
    ("pfv_gop_encoder_encode_pframe_dev", c_int, [_P, _P]),
    ("pfv_gop_encoder_encode_dropframe", c_int, [_P]),
    ("pfv_gop_encoder_flush", c_int, [_P]),
    ("pfv_gop_encoder_finish", c_int, [_P]),
    ("pfv_gop_encoder_drain", c_int, [_P, POINTER(_P), POINTER(c_size_t)]),
    ("pfv_gop_encoder_bytes", c_int, [_P, POINTER(_P), POINTER(c_size_t)]),
    ("pfv_gop_encoder_drain_iov", c_int, [_P, POINTER(_P), POINTER(c_size_t)]),
    ("pfv_gop_encoder_batches", ctypes.c_long, [_P]),
    ("pfv_gop_encoder_stats", c_int, [_P, _P, c_int]),
    ("pfv_gop_encoder_destroy", None, [_P]),
    ("pfv_gop_decoder_create", c_int, [_P, _P, c_size_t, c_int, c_int, c_int, POINTER(_P)]),
    ("pfv_gop_decoder_width", c_int, [_P]),
    ("pfv_gop_decoder_height", c_int, [_P]),
    ("pfv_gop_decoder_framerate", c_int, [_P]),
    ("pfv_gop_decoder_batches", ctypes.c_long, [_P]),
    ("pfv_gop_decoder_stats", c_int, [_P, _P, c_int]),
    ("pfv_gop_decoder_set_output_device", c_int, [_P, c_int]),
    ("pfv_decoder_entropy_counts", None, [_P, _P]),
    ("pfv_decoder_set_output_device", c_int, [_P, c_int]),
    ("pfv_batch_decoder_entropy_counts", None, [_P, _P]),
    ("pfv_gop_decoder_reset", c_int, [_P]),
    ("pfv_gop_decoder_advance_frame", c_int, [_P, _P, _P]),
    ("pfv_gop_decoder_advance_delta", c_int, [_P, ctypes.c_double, _P, _P]),
    ("pfv_gop_decoder_destroy", None, [_P]),
    ("pfv_serialize_iframe_payload", c_size_t, [_P, c_int, _P, c_size_t]),
    ("pfv_serialize_pframe_payload", c_size_t, [_P, _P, _P, c_int, _P, c_size_t]),
    ("pfv_parse_iframe_payload", c_int, [_P, c_size_t, c_int, c_int, _P, _P]),
    ("pfv_parse_pframe_payload", c_int, [_P, c_size_t, c_int, c_int, _P, _P, _P, _P]),
    ("pfv_parse_payload_sparse", c_int, [c_int, _P, c_size_t, c_int, c_int, _P, _P, _P, _P, c_size_t, POINTER(c_size_t), _P]),
    ("pfv_decoder_create", c_int, [_P, _P, c_size_t, POINTER(_P)]),
    ("pfv_decoder_destroy", None, [_P]),
    ("pfv_decoder_set_lookahead", c_int, [_P, c_int]),
    ("pfv_decoder_width", c_int, [_P]),
    ("pfv_decoder_height", c_int, [_P]),
    ("pfv_decoder_framerate", c_int, [_P]),
    ("pfv_decoder_reset", c_int, [_P]),
    ("pfv_decoder_advance_frame", c_int, [_P, _P, _P]),
    ("pfv_decoder_advance_delta", c_int, [_P, ctypes.c_double, _P, _P]),
]

_lib = None
_lib_path = None


def lib_path() -> str:
    return DEFAULT_LIB


def load():
    """Load the shared object and bind every symbol; raises if it is missing (no fallback)."""
    global _lib, _lib_path
    path = lib_path()
    if _lib is not None and _lib_path == path:
        return _lib
    if not os.path.exists(path):
        raise PfvError(PFV_ERR_NO_DEVICE, f"{path} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = ctypes.CDLL(path)
    for name, restype, argtypes in SIGNATURES:
        fn = getattr(lib, name)   # AttributeError if the .so does not export a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    _lib, _lib_path = lib, path
    return lib


def check(ctx_handle, rc: int):
    if rc != PFV_OK:
        msg = load().pfv_last_error(ctx_handle)
        raise PfvError(rc, msg.decode() if msg else "")
