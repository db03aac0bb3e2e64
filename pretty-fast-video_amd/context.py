"""Context = the parallel resource handed to every operator.

Takes the place of the reference's ``tp: &rayon::ThreadPool`` argument (src/common.rs:351)
and of ``num_threads`` in ``Encoder::new`` / ``Decoder::new`` (src/enc.rs:37, src/dec.rs:38):
here the resource is one MI355X device + one HIP stream.
"""
from __future__ import annotations

import ctypes
import weakref

import numpy as np

from . import _lib


class Context:
    def __init__(self, device: int = 0, priority: int = 0):
        """priority: > 0 / < 0 = the device's greatest / least HIP stream priority for this context's stream (pfv_ctx_create_prio)"""
        self._lib = _lib.load()
        h = ctypes.c_void_p()
        rc = (self._lib.pfv_ctx_create_prio(int(device), int(priority), ctypes.byref(h)) if priority else
              self._lib.pfv_ctx_create(int(device), ctypes.byref(h)))
        if rc != _lib.PFV_OK:
            msg = self._lib.pfv_last_error(None)
            raise _lib.PfvError(rc, msg.decode() if msg else "")
        self.handle = h
        self.device = int(device)
        self._sessions = weakref.WeakSet()   # sessions die with their context (they hold device buffers of it)
        self._pinned = []                    # page-locked host blocks handed out by host_array()
        self.keep_alive = False              # a library call is stuck inside this context on another thread (comm.py's init watchdog):
        #                                      destroying it would pull the context from under that call

    def pci_bus_id(self) -> str:
        """PCI address of the context's device ("domain:bus:device.function")"""
        buf = ctypes.create_string_buffer(32)
        self.check(self._lib.pfv_ctx_pci_bus_id(self.handle, buf, 32))
        return buf.value.decode()

    # -- lifetime
    def close(self):
        if getattr(self, "handle", None):
            if self.keep_alive:
                return                       # leaked on purpose; the process is expected to leave through os._exit
            for s in list(self._sessions):
                s.close()
            for p in self._pinned:
                self._lib.pfv_host_free(self.handle, ctypes.c_void_p(p))
            self._pinned = []
            self._lib.pfv_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- helpers
    def check(self, rc: int):
        _lib.check(self.handle, rc)

    def sync(self):
        self.check(self._lib.pfv_ctx_sync(self.handle))

    def set_option(self, option: int, value: int):
        """pfv_ctx_set_option: applies to plane-level operators on this context and to sessions created afterwards"""
        self.check(self._lib.pfv_ctx_set_option(self.handle, int(option), int(value)))

    def get_option(self, option: int) -> int:
        v = ctypes.c_int()
        self.check(self._lib.pfv_ctx_get_option(self.handle, int(option), ctypes.byref(v)))
        return int(v.value)

    # timing / ordering events on this context's stream (pfv_event_*)
    def event(self):
        h = ctypes.c_void_p()
        self.check(self._lib.pfv_event_create(self.handle, ctypes.byref(h)))
        return h

    def record(self, ev):
        self.check(self._lib.pfv_event_record(ev))

    def wait_event(self, ev):
        """this context's stream waits on the device for an event recorded on another context's stream"""
        self.check(self._lib.pfv_ctx_wait_event(self.handle, ev))

    def event_destroy(self, ev):
        self._lib.pfv_event_destroy(ev)

    def device_sync(self):
        """hipDeviceSynchronize: every stream of the device"""
        self.check(self._lib.pfv_device_sync(self.handle))

    @property
    def stream(self) -> int:
        return int(self._lib.pfv_ctx_stream(self.handle) or 0)

    def alloc(self, nbytes: int) -> int:
        p = ctypes.c_void_p()
        self.check(self._lib.pfv_dev_alloc(self.handle, int(nbytes), ctypes.byref(p)))
        return int(p.value)

    def free(self, ptr: int):
        self.check(self._lib.pfv_dev_free(self.handle, ctypes.c_void_p(ptr)))

    def synth_frames_dev(self, width: int, height: int, seeds, t: int, frames_dev: int, kind="pan"):
        """frame t of len(seeds) synthetic streams (synth.SyntheticStream bytes) written on the device, asynchronously on
        the context's stream, into frames_dev (len(seeds) packed Y|U|V frames back to back); kind: "pan" | "low_motion" | "static" """
        sd = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint64))
        k = {"pan": 0, "low_motion": 1, "static": 2}[kind]
        if k == 0:
            self.check(self._lib.pfv_synth_frames_dev(self.handle, int(width), int(height), int(sd.size), ptr(sd), int(t), ctypes.c_void_p(int(frames_dev))))
            return
        self.check(self._lib.pfv_synth_frames_kind_dev(self.handle, int(width), int(height), int(sd.size), ptr(sd), int(t), k,
                                                       ctypes.c_void_p(int(frames_dev))))

    def host_array(self, nbytes: int) -> np.ndarray:
        """uint8 array over page-locked host memory (freed when the context closes)"""
        p = ctypes.c_void_p()
        self.check(self._lib.pfv_host_alloc(self.handle, int(nbytes), ctypes.byref(p)))
        self._pinned.append(p.value)
        return np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(int(nbytes),))

    def host_free(self, arr: np.ndarray):
        """give a host_array() block back before the context closes (the array must not be used afterwards)"""
        p = arr.ctypes.data
        if p in self._pinned:
            self._pinned.remove(p)
            self.check(self._lib.pfv_host_free(self.handle, ctypes.c_void_p(p)))

    def upload(self, dst_dev: int, src: np.ndarray):
        src = np.ascontiguousarray(src)
        self.check(self._lib.pfv_dev_upload(self.handle, ctypes.c_void_p(dst_dev), src.ctypes.data_as(ctypes.c_void_p),
                                            src.nbytes))

    def download(self, dst: np.ndarray, src_dev: int):
        assert dst.flags["C_CONTIGUOUS"]
        self.check(self._lib.pfv_dev_download(self.handle, dst.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(src_dev),
                                              dst.nbytes))


class Graph:
    """A recorded sequence of ``*_dev`` calls (HIP graph): ``with Graph(ctx) as g: ...dev calls...`` records, ``g.launch()``
    replays them as one launch.  See include/pfv_hip.h (pfv_graph_begin) for the ping-pong rule."""

    def __init__(self, ctx: Context):
        self.ctx, self.handle = ctx, None

    def __enter__(self):
        self.ctx.check(self.ctx._lib.pfv_graph_begin(self.ctx.handle))
        return self

    def __exit__(self, exc_type, exc, tb):
        h = ctypes.c_void_p()
        rc = self.ctx._lib.pfv_graph_end(self.ctx.handle, ctypes.byref(h))
        if exc_type is None:
            self.ctx.check(rc)
            self.handle = h
        elif rc == _lib.PFV_OK:
            self.ctx._lib.pfv_graph_destroy(h)
        return False

    def launch(self):
        self.ctx.check(self.ctx._lib.pfv_graph_launch(self.handle))

    def close(self):
        if self.handle is not None and self.ctx.handle:
            self.ctx._lib.pfv_graph_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ptr(a: np.ndarray) -> ctypes.c_void_p:
    return a.ctypes.data_as(ctypes.c_void_p)
