"""A host-memory stand-in for xclim_amd._capi.Device: every compute entry point of the C ABI is a no-op that returns
XH_OK, memory entry points work on host buffers.  ONLY for plumbing tests of multi-rank launches on machines without a
GPU (tests/test_shard_gloo.py drives ``bench.py --gpus 2`` through it: environment handling, rendezvous fallback, barriers,
max-over-ranks timing, the one JSON line).  Numbers produced with it mean nothing and are labelled ``"data": "mock"``."""
import ctypes as C
import threading

import numpy as np

from xclim_amd._capi import Device, DeviceArray


class _MockLib:
    def __getattr__(self, name):
        if not name.startswith("xh_"):
            raise AttributeError(name)

        def call(*args):
            return _MockLib._SPECIAL.get(name, lambda *a: 0)(*args)

        return call

    @staticmethod
    def _val(p):
        return p.value if hasattr(p, "value") else int(p or 0)

    @staticmethod
    def _copy(ctx, dst, src, n):
        if n:
            C.memmove(_MockLib._val(dst), _MockLib._val(src), int(n))
        return 0

    @staticmethod
    def _copy2d(ctx, dst, dpitch, src, spitch, width, height, kind, lane, blocking):
        d, s = _MockLib._val(dst), _MockLib._val(src)
        for r in range(int(height)):
            C.memmove(d + r * int(dpitch), s + r * int(spitch), int(width))
        return 0

    @staticmethod
    def _memset(ctx, p, v, n):
        C.memset(_MockLib._val(p), int(v), int(n))
        return 0

    @staticmethod
    def _timer_stop(ctx, ms):
        ms._obj.value = 1.0
        return 0

    @staticmethod
    def _name(ctx, buf, n):
        buf.value = b"mock device (no GPU)"
        return 0


_MockLib._SPECIAL = {"xh_memcpy_d2h": _MockLib._copy, "xh_memcpy_h2d": _MockLib._copy, "xh_memcpy_d2d": _MockLib._copy,
                     "xh_memcpy2d": _MockLib._copy2d, "xh_memset": _MockLib._memset, "xh_timer_stop": _MockLib._timer_stop,
                     "xh_device_name": _MockLib._name}


class MockDevice(Device):
    def __init__(self, device: int = 0):  # noqa: D107 — no library, no context
        self.lib, self.ctx, self.index = _MockLib(), C.c_void_p(1), device
        self.lock, self.trace = threading.RLock(), None
        self._pool, self._pool_bytes, self._pool_cap, self._pinned = {}, 0, 0, {}
        self._bufs = {}
        self._inputs, self._inputs_bytes, self._inputs_cap, self._inputs_min = {}, 0, 0, 1 << 62   # (no input cache)
        self._inputs_mode, self._keep_depth, self._forget_hooks = "scope", 0, []   # (scopes exist, no field is ever large enough)

    def empty(self, shape, dtype) -> DeviceArray:
        shape = (shape,) if np.isscalar(shape) else tuple(shape)
        nbytes = max(int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize, 16)
        buf = np.zeros(nbytes, np.uint8)
        self._bufs[buf.ctypes.data] = buf
        return DeviceArray(self, buf.ctypes.data, shape, dtype, alloc=nbytes)

    def _release(self, ptr: int, nbytes: int) -> None:
        self._bufs.pop(ptr, None)

    def close(self):
        self._bufs.clear()
