"""ctypes binding of libxclimhip.so (include/xclim_hip.h) — the thin C-ABI layer named by the north star.

No torch, no cffi (not installed offline).  The library is looked up in-tree (xclim_amd/lib/libxclimhip.so, built by
``__graft_entry__.build()`` / ``make -C xclim_amd/csrc``).  Every compute call goes to the HIP kernels; there is NO CPU
fallback: if the library or a GPU is missing the call raises :class:`BackendUnavailable`.
"""

from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libxclimhip.so")

XH_OK = 0
XH_ERR_HIP, XH_ERR_ARG, XH_ERR_LAYOUT, XH_ERR_OP, XH_ERR_NOTIMPL, XH_ERR_NODEVICE, XH_ERR_LIMIT = -1, -2, -3, -4, -5, -6, -7

OPS = {">": 0, "gt": 0, "<": 1, "lt": 1, ">=": 2, "ge": 2, "<=": 3, "le": 3, "==": 4, "eq": 4, "!=": 5, "ne": 5}
THR_SCALAR_F32, THR_SCALAR_F64, THR_DOY_F64, THR_DOY_F32, THR_FULL_F64, THR_FULL_F32 = range(6)
REDUCERS = {"sum": 0, "integral": 0, "mean": 1, "min": 2, "max": 3, "std": 4, "var": 5, "count": 6, "argmin": 7, "argmax": 8}
RUN_STATS = {"max": 0, "min": 1, "sum": 2, "count": 3, "mean": 4, "std": 5, "first": 6, "last": 7, "plainsum": 8}


class PrecisionWarning(UserWarning):
    """A float64 input was rounded to float32: the kernels compute on float32 fields (xclim keeps the input dtype, so
    values differ at ~1e-7 relative and comparisons of values that close to a threshold can flip)."""


def warn_downcast(a, what: str) -> None:
    """One PrecisionWarning per call site when `a` is a float64 array (python floats and float32 pass silently)."""
    import warnings

    if getattr(a, "dtype", None) == np.float64 and getattr(a, "ndim", 0) > 0:
        warnings.warn(f"{what}: float64 input is rounded to float32 (the HIP kernels compute on float32 fields)",
                      PrecisionWarning, stacklevel=3)


class Float64FieldError(TypeError):
    """A float64 FIELD reached an entry point that only exists for float32 fields.  The reference computes in the input
    dtype; rounding the field first can flip a count next to a threshold, so it is refused unless asked for."""


def handle_float64(a, what: str) -> None:
    """Policy for float64 fields on float32-only kernels (``XCLIM_AMD_FLOAT64``): "raise" (default) -> Float64FieldError;
    "round" -> PrecisionWarning and the field is rounded to float32.  threshold_count / count_occurrences /
    select_resample_op / calc_perc have native float64 kernels (xh_*_f64) and never come here."""
    if getattr(a, "dtype", None) != np.float64 or getattr(a, "ndim", 0) == 0:
        return
    if os.environ.get("XCLIM_AMD_FLOAT64", "raise").lower() == "round":
        warn_downcast(a, what)
        return
    raise Float64FieldError(f"{what}: float64 fields are only served by threshold_count, count_occurrences, select_resample_op and "
                            "calc_perc (xh_*_f64); cast to float32 yourself, or set XCLIM_AMD_FLOAT64=round to have it "
                            "rounded with a PrecisionWarning")


class BackendUnavailable(RuntimeError):
    """libxclimhip.so is not built/loadable or no MI355X device is visible."""


class XclimHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[xclimhip {code}] {msg}")
        self.code = code
        self.msg = msg


_i32, _i64, _u32, _u64 = C.c_int32, C.c_int64, C.c_uint32, C.c_uint64
_vp, _dbl, _flt, _int, _sz = C.c_void_p, C.c_double, C.c_float, C.c_int, C.c_size_t

# name -> argtypes (restype is always int unless listed in _RESTYPES); mirrors include/xclim_hip.h one to one
SIGNATURES: dict[str, list] = {
    "xh_abi_version": [],
    "xh_last_error": [],
    "xh_device_count": [C.POINTER(_int)],
    "xh_create": [_int, C.POINTER(_vp)],
    "xh_destroy": [_vp],
    "xh_sync": [_vp],
    "xh_device_name": [_vp, C.c_char_p, _sz],
    "xh_mem_info": [_vp, C.POINTER(_sz), C.POINTER(_sz)],
    "xh_malloc": [_vp, _sz, C.POINTER(_vp)],
    "xh_free": [_vp, _vp],
    "xh_memset": [_vp, _vp, _int, _sz],
    "xh_memcpy_h2d": [_vp, _vp, _vp, _sz],
    "xh_memcpy_d2h": [_vp, _vp, _vp, _sz],
    "xh_memcpy_d2d": [_vp, _vp, _vp, _sz],
    "xh_host_alloc": [_vp, _sz, C.POINTER(_vp)],
    "xh_host_free": [_vp, _vp],
    "xh_host_register": [_vp, _vp, _sz],
    "xh_host_unregister": [_vp, _vp],
    "xh_memcpy2d": [_vp, _vp, _sz, _vp, _sz, _sz, _sz, _int, _int, _int],
    "xh_lane_fence": [_vp, _int, _int],
    "xh_lane_sync": [_vp, _int],
    "xh_timer_start": [_vp],
    "xh_timer_stop": [_vp, C.POINTER(_flt)],
    "xh_stream": [_vp, C.POINTER(_vp)],
    "xh_comm_unique_id": [_vp],
    "xh_comm_init": [_vp, _int, _int, _vp, C.POINTER(_vp)],
    "xh_comm_destroy": [_vp],
    "xh_comm_size": [_vp, C.POINTER(_int), C.POINTER(_int)],
    "xh_comm_allgather": [_vp, _vp, _vp, _sz, _int],
    "xh_comm_fence": [_vp, _int],
    "xh_comm_sync": [_vp],
    "xh_comm_allreduce_f64": [_vp, C.POINTER(_dbl), _int, _int],
    "xh_comm_barrier": [_vp],
    "xh_fill_synthetic": [_vp, _vp, _i64, _i64, _i64, _int, _u64, _i64, _vp, _flt, _flt, _u32],
    "xh_transpose_f32": [_vp, _vp, _i64, _i64, _i64, _vp, _i64],
    "xh_threshold_count": [_vp, _vp, _i64, _i64, _i64, _i64, _int, _int, _dbl, _vp, _i64, _vp, _vp, _int, _vp, _vp],
    "xh_domain_count": [_vp, _vp, _i64, _i64, _i64, _i64, _int, _dbl, _int, _dbl, _int, _vp, _int, _vp, _vp],
    "xh_resample_reduce": [_vp, _vp, _i64, _i64, _i64, _i64, _int, _int, _vp, _int, _vp, _vp],
    "xh_apply_missing_mask": [_vp, _vp, _int, _vp, _vp, _int, _i64, _vp],
    "xh_rolling_reduce": [_vp, _vp, _i64, _i64, _i64, _i64, _int, _int, _int, _vp, _i64],
    "xh_cumsum_reset": [_vp, _vp, _i64, _i64, _i64, _i64, _int, _vp, _i64],
    "xh_rle": [_vp, _vp, _i64, _i64, _i64, _i64, _int, _vp, _i64],
    "xh_run_stats": [_vp, _vp, _i64, _i64, _i64, _i64, _int, _dbl, _int, _int, _int, _vp, _int, _int, _vp, _vp],
    "xh_bivariate_count": [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _int, _dbl, _int, _dbl, _int, _vp, _int, _vp, _vp],
    "xh_range_reduce": [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _int, _int, _vp, _int, _vp, _vp],
    "xh_compare_map": [_vp, _vp, _i64, _i64, _i64, _int, _dbl, _int, _vp, _i64, _int, _vp, _i64],
    "xh_thresholded_reduce": [_vp, _vp, _i64, _i64, _i64, _i64, _int, _dbl, _int, _int, _vp, _int, _vp, _vp],
    "xh_mask_rows": [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _int, _vp, _vp, _int, _vp, _i64],
    "xh_doy_mean_std": [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _int, _int, _int, _vp, _vp],
    "xh_spell_mask": [_vp, _vp, _i64, _i64, _i64, _i64, _int, _int, _int, _dbl, _vp, _vp, _i64],
    "xh_percentile_doy_count": [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _int, _int, _int, _dbl, _dbl, _dbl, _int, _vp, _int, _vp, _vp],
    "xh_doy_broadcast": [_vp, _vp, _int, _i64, _vp, _i64, _vp],
    "xh_within_bnds_doy": [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _int, _vp, _vp],
    "xh_compare_doy": [_vp, _vp, _i64, _i64, _i64, _i64, _int, _vp, _int, _vp, _vp, _i64],
    "xh_select_rows": [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _i64, _vp, _i64],
    "xh_mask_u8_to_f32": [_vp, _vp, _i64, _vp],
    "xh_run_stats_doy": [_vp, _vp, _i64, _i64, _i64, _i64, _int, _vp, _int, _vp, _int, _int, _vp, _int, _vp, _vp],
    "xh_precip_over_doy": [_vp, _vp, _i64, _i64, _i64, _i64, _int, _dbl, _vp, _int, _vp, _vp, _int, _vp, _vp, _vp],
    "xh_spell_run_stats": [_vp, _vp, _i64, _i64, _i64, _i64, _int, _int, _int, _dbl, _vp, _int, _vp, _int, _vp, _vp],
    "xh_spell_mask_multi": [_vp, _vp, _int, _vp, _int, _i64, _i64, _i64, _i64, _int, _int, _int, _vp, _vp, _i64],
    "xh_runs_with_holes": [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _int, _int, _vp, _i64],
    "xh_run_events": [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _int, _int, _vp, _vp, _vp, _vp, _vp],
    "xh_suspicious_run": [_vp, _vp, _i64, _i64, _i64, _i64, _int, _int, _dbl, _vp, _i64],
    "xh_keep_longest_run": [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _int, _vp, _i64],
    "xh_season": [_vp, _vp, _i64, _i64, _i64, _i64, _int, _vp, _vp, _int, _vp, _vp, _vp],
    "xh_max_run_sum": [_vp, _vp, _i64, _i64, _i64, _i64, _int, _vp, _int, _int, _vp],
    "xh_nan_quantile": [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _int, _dbl, _dbl, _vp],
    "xh_threshold_count_doy": [_vp, _vp, _i64, _i64, _i64, _i64, _int, _vp, _i64, _int, _vp, _vp, _int, _vp, _vp],
    "xh_threshold_count_f64": [_vp, _vp, _i64, _i64, _i64, _i64, _int, _int, _dbl, _vp, _i64, _vp, _vp, _int, _vp, _vp],
    "xh_resample_reduce_f64": [_vp, _vp, _i64, _i64, _i64, _i64, _int, _int, _vp, _int, _vp, _vp],
    "xh_nan_quantile_f64": [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _int, _dbl, _dbl, _vp],
    "xh_weighted_quantile": [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _int, _vp],
    "xh_percentile_doy": [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _int, _int, _int, _vp, _int, _dbl, _dbl, _vp],
    "xh_percentile_doy_mapped": [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _int, _int, _int, _vp, _int, _dbl, _dbl, _vp, _i64, _vp],
    "xh_doy_interp": [_vp, _vp, _int, _i64, _vp, _vp, _vp, _vp, _int, _vp, _vp],
    "xh_quantile_series": [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _int, _vp],
    "xh_eqm_train": [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _int, _int, _vp, _vp],
    "xh_eqm_train_window": [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _int, _vp, _vp, _int, _int, _vp, _int, _int, _vp, _vp],
    "xh_dqm_train_window": [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _int, _vp, _vp, _int, _int, _vp, _int, _int, _vp, _vp, _vp, _vp],
    "xh_eqm_train_groups": [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _int, _vp, _int, _int, _vp, _vp],
    "xh_dqm_train_groups": [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _int, _vp, _int, _int, _vp, _vp, _vp, _vp],
    "xh_eqm_adjust": [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _int, _int, _int, _int, _vp, _i64],
    "xh_eqm_adjust_g2d": [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _int, _int, _int, _int, _int, _vp, _i64],
    "xh_apply_factor": [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _int, _vp, _i64],
    "xh_plane_nearest": [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _vp, _i64],
    "xh_plane_linear": [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _int, _int, _int, _vp, _i64],
    "xh_qdm_adjust": [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _int, _int, _int, _int, _vp],
    "xh_qdm_adjust_groups": [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _int, _vp, _vp, _int, _int, _int, _int, _vp, _i64],
    "xh_quantile_cells": [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp],
    "xh_adapt_freq": [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _dbl, _u64, _vp, _i64, _vp],
    "xh_mask_doy_cells": [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _i64],
    "xh_rolling_dot": [_vp, _vp, _i64, _i64, _i64, _i64, _int, _vp, _vp, _i64],
    "xh_mask_days_cells": [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _int, _vp, _vp, _vp, _i64],
    "xh_poly_trend": [_vp, _vp, _i64, _i64, _i64, _i64, _int, _vp, _vp, _vp],
    "xh_trend_apply": [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _int, _vp, _i64],
    "xh_poly_trend_u": [_vp, _vp, _i64, _i64, _i64, _i64, _int, _vp, _vp, _vp, _vp],
    "xh_trend_apply_u": [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _int, _vp, _i64],
    "xh_window_nanmean": [_vp, _vp, _i64, _i64, _i64, _i64, _int, _vp, _i64],
    "xh_poly_trend_groups": [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _int, _vp, _int, _vp, _vp],
    "xh_trend_apply_groups": [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _int, _vp, _vp, _vp, _int, _vp, _i64],
}
_RESTYPES = {"xh_last_error": C.c_char_p}

_lib = None
_lib_lock = threading.Lock()


def library_path() -> str:
    return _LIB_PATH


def load_library() -> C.CDLL:
    """Load libxclimhip.so and declare every prototype of include/xclim_hip.h.  Raises BackendUnavailable."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(_LIB_PATH):
            raise BackendUnavailable(
                f"{_LIB_PATH} not found — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C xclim_amd/csrc` (needs hipcc)"
            )
        try:
            lib = C.CDLL(_LIB_PATH)
        except OSError as err:  # pragma: no cover
            raise BackendUnavailable(f"cannot load {_LIB_PATH}: {err}") from err
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the header and the library diverge
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, _int)
        _lib = lib
        return lib


def _check(lib, rc: int):
    if rc != XH_OK:
        msg = lib.xh_last_error()
        msg = msg.decode() if msg else ""
        if rc == XH_ERR_NODEVICE:
            raise BackendUnavailable(msg)
        if rc == XH_ERR_OP:
            raise ValueError(msg)
        raise XclimHipError(rc, msg)


def device_count() -> int:
    lib = load_library()
    n = _int(0)
    _check(lib, lib.xh_device_count(C.byref(n)))
    return n.value


class DeviceArray:
    """A typed device buffer owned by a :class:`Device` (hipMalloc'd through xh_malloc)."""

    __slots__ = ("dev", "ptr", "shape", "dtype", "_owner", "_alloc")

    def __init__(self, dev: "Device", ptr: int, shape, dtype, owner=True, alloc: int = 0):
        self.dev = dev
        self.ptr = ptr
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self._owner = owner
        self._alloc = int(alloc)  # bytes obtained from xh_malloc (0: foreign / view); lets free() hand it to the pool

    @property
    def size(self) -> int:
        return int(np.prod(self.shape, dtype=np.int64)) if self.shape else 1

    @property
    def nbytes(self) -> int:
        return self.size * self.dtype.itemsize

    def reshape(self, *shape) -> "DeviceArray":
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        shape = tuple(shape)
        if -1 in shape:
            known = int(np.prod([s for s in shape if s != -1], dtype=np.int64))
            shape = tuple(self.size // max(known, 1) if s == -1 else s for s in shape)
        assert int(np.prod(shape, dtype=np.int64)) == self.size, (shape, self.shape)
        view = DeviceArray(self.dev, self.ptr, shape, self.dtype, owner=False)
        view._owner = self  # keep the parent alive
        return view

    def get(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=self.dtype)
        if out.size:
            lib = self.dev.lib
            if self.dev.trace is not None:
                self.dev.trace.append(("d2h", (out.nbytes,)))
            _check(lib, lib.xh_memcpy_d2h(self.dev.ctx, out.ctypes.data_as(_vp), _vp(self.ptr), out.nbytes))
        return out

    def free(self):
        if self._owner is True and self.ptr and self.dev is not None and self.dev.ctx:
            self.dev._release(self.ptr, self._alloc)
        self.ptr = 0
        self._owner = False

    def __del__(self):
        try:
            self.free()
        except Exception:  # pragma: no cover
            pass


class Device:
    """One xh_ctx (HIP stream + scratch) on one GPU.  Not re-entrant: calls are serialised by a lock."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        ctx = _vp()
        _check(self.lib, self.lib.xh_create(device, C.byref(ctx)))
        self.ctx = ctx
        self.index = device
        self.lock = threading.RLock()
        self.trace = None  # list of (entry point, args) while start_trace() is active
        # Freed buffers are kept per exact size and handed out again: hipMalloc / hipFree cost ~0.1-0.4 ms each and
        # hipFree synchronises the device, which is as long as a whole kernel of this library.  Re-use is safe because
        # every kernel of a context runs on its one stream (stream order protects a buffer that is still being read).
        self._pool: dict[int, list[tuple[int, int]]] = {}   # size -> [(release number, address)], oldest first
        self._pool_seq = 0
        self._pool_bytes = 0
        # cap: XCLIM_AMD_POOL_BYTES, else a quarter of the device's memory (72 GB of an MI355X's 288; at least 32 GiB) — a grouped
        # DQM adjust of a 30-year 1440 x 90 band turns over four 5.7 GB buffers per call, and giving those back to the driver
        # costs two seconds per call (hipFree + hipMalloc of multi-GB ranges) against 27 ms of kernels.  A failed allocation
        # empties the pool and retries (empty()).
        cap = os.environ.get("XCLIM_AMD_POOL_BYTES")
        self._pool_cap = int(cap) if cap is not None else None  # None: decided at the first release
        self._pinned: dict[int, int] = {}  # page-locked host ranges we own or registered: address -> bytes
        # device copies of large host inputs, recognised again across calls (resident())
        self._inputs: dict = {}
        self._inputs_bytes = 0
        # "scope" (default): only inside `with dev.keep_inputs():` (patch.install() opens one around every Indicator
        # call); "always": across every call of the process (the caller promises not to edit fields in place, or calls
        # forget_inputs() after doing so); "off"
        self._inputs_mode = os.environ.get("XCLIM_AMD_INPUT_CACHE", "scope").lower()
        if self._inputs_mode not in ("scope", "always", "off"):
            raise ValueError(f"XCLIM_AMD_INPUT_CACHE must be scope, always or off, got {self._inputs_mode!r}")
        self._keep_depth = 0
        self._forget_hooks: list = []  # run by forget_inputs(): caches that live and die with the input copies (xr_adapter's tables)
        cap = os.environ.get("XCLIM_AMD_INPUT_CACHE_BYTES")
        self._inputs_cap = int(cap) if cap is not None else None  # None: half of the free device memory at first use
        self._inputs_min = int(os.environ.get("XCLIM_AMD_INPUT_CACHE_MIN", str(32 << 20)))

    # ---- memory ----
    def empty(self, shape, dtype) -> DeviceArray:
        shape = (shape,) if np.isscalar(shape) else tuple(shape)
        nbytes = max(int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize, 16)
        with self.lock:
            free = self._pool.get(nbytes)
            if free:
                self._pool_bytes -= nbytes
                _, ptr = free.pop()        # (the most recently released one)
                if not free:
                    del self._pool[nbytes]
                return DeviceArray(self, ptr, shape, dtype, alloc=nbytes)
        p = _vp()
        rc = self.lib.xh_malloc(self.ctx, nbytes, C.byref(p))
        if rc != XH_OK and self._pool_bytes:
            self.trim()  # out of memory with buffers parked in the pool: give them back and retry once
            rc = self.lib.xh_malloc(self.ctx, nbytes, C.byref(p))
        while rc != XH_OK and self._inputs:
            # ... or with remembered input copies (resident()): the least recently used one goes, its memory passes
            # through the pool back to the driver, and the allocation is tried again
            self._drop_input(next(iter(self._inputs)))
            self.trim()
            rc = self.lib.xh_malloc(self.ctx, nbytes, C.byref(p))
        _check(self.lib, rc)
        return DeviceArray(self, p.value, shape, dtype, alloc=nbytes)

    def _release(self, ptr: int, nbytes: int) -> None:
        """A freed buffer is parked (per exact size).  Over the cap, or with more than 8 of one size, the OLDEST parked buffers go
        back to the driver — of any size: the pool follows the working set (a bench leg on 45 GB fields must not leave the pool
        full of sizes nobody asks for again while the next leg's 5.7 GB buffers go through hipFree / hipMalloc on every call)."""
        victims = []
        with self.lock:
            if self._pool_cap is None:
                try:
                    self._pool_cap = max(32 << 30, self.mem_info()[1] // 4)
                except Exception:
                    self._pool_cap = 32 << 30
            if not nbytes or nbytes > self._pool_cap:
                victims.append(ptr)
            else:
                self._pool_seq = getattr(self, "_pool_seq", 0) + 1
                self._pool.setdefault(nbytes, []).append((self._pool_seq, ptr))
                self._pool_bytes += nbytes
                while True:
                    if len(self._pool.get(nbytes, ())) > 8:
                        size = nbytes
                    elif self._pool_bytes > self._pool_cap:
                        size = min((lst[0][0], sz) for sz, lst in self._pool.items())[1]
                    else:
                        break
                    _, p = self._pool[size].pop(0)
                    if not self._pool[size]:
                        del self._pool[size]
                    self._pool_bytes -= size
                    victims.append(p)
        for p in victims:
            self.lib.xh_free(self.ctx, _vp(p))

    def trim(self) -> None:
        """Return every pooled buffer to the driver."""
        with self.lock:
            pool, self._pool, self._pool_bytes = self._pool, {}, 0
        for ptrs in pool.values():
            for _, ptr in ptrs:
                self.lib.xh_free(self.ctx, _vp(ptr))

    # ---- pinned host memory + copy lanes (block adapter, blocks.py) ----
    def pinned_empty(self, shape, dtype) -> np.ndarray:
        """numpy array over page-locked host memory (xh_host_alloc): strided copies from / to it are asynchronous and run
        at the full PCIe rate.  Freed when the array (and every view of it) is garbage collected."""
        import weakref

        shape = (shape,) if np.isscalar(shape) else tuple(shape)
        nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
        p = _vp()
        _check(self.lib, self.lib.xh_host_alloc(self.ctx, max(nbytes, 16), C.byref(p)))
        buf = (C.c_char * max(nbytes, 16)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape, dtype=np.int64))).reshape(shape)
        self._pinned[p.value] = max(nbytes, 16)
        lib, ctx, pinned, addr = self.lib, self.ctx, self._pinned, p.value

        def _free():
            pinned.pop(addr, None)
            if ctx:
                lib.xh_host_free(ctx, _vp(addr))

        weakref.finalize(buf, _free)
        return arr

    def register(self, arr: np.ndarray) -> None:
        """Page-lock caller memory (hipHostRegister) for the lifetime of the registration; costs about as much as one
        copy of the array, so it pays for inputs that are streamed more than once."""
        assert arr.flags.c_contiguous
        _check(self.lib, self.lib.xh_host_register(self.ctx, _vp(arr.ctypes.data), arr.nbytes))
        self._pinned[arr.ctypes.data] = arr.nbytes

    def unregister(self, arr: np.ndarray) -> None:
        if self._pinned.pop(arr.ctypes.data, None) is not None:
            _check(self.lib, self.lib.xh_host_unregister(self.ctx, _vp(arr.ctypes.data)))

    def is_pinned(self, arr: np.ndarray) -> bool:
        a0, a1 = arr.ctypes.data, arr.ctypes.data + arr.nbytes
        return any(p <= a0 and a1 <= p + n for p, n in self._pinned.items())

    def copy2d(self, dst: int, dpitch: int, src: int, spitch: int, width: int, height: int, kind: str, lane: int = 0,
               blocking: bool = True) -> None:
        """xh_memcpy2d: `height` rows of `width` bytes; kind "h2d" | "d2h" | "d2d"; lanes 0 compute / 1 copy-in / 2 copy-out."""
        _check(self.lib, self.lib.xh_memcpy2d(self.ctx, _vp(dst), dpitch, _vp(src), spitch, width, height,
                                              {"h2d": 0, "d2h": 1, "d2d": 2}[kind], lane, int(bool(blocking))))

    def lane_fence(self, from_lane: int, to_lane: int) -> None:
        _check(self.lib, self.lib.xh_lane_fence(self.ctx, from_lane, to_lane))

    def lane_sync(self, lane: int) -> None:
        _check(self.lib, self.lib.xh_lane_sync(self.ctx, lane))

    def zeros(self, shape, dtype) -> DeviceArray:
        a = self.empty(shape, dtype)
        if a.nbytes:
            _check(self.lib, self.lib.xh_memset(self.ctx, _vp(a.ptr), 0, a.nbytes))
        return a

    def copy_d2d(self, dst_ptr: int, src_ptr: int, nbytes: int) -> None:
        _check(self.lib, self.lib.xh_memcpy_d2d(self.ctx, _vp(dst_ptr), _vp(src_ptr), int(nbytes)))

    def to_device(self, arr: np.ndarray, dtype=None) -> DeviceArray:
        arr = np.ascontiguousarray(arr, dtype=dtype)
        d = self.empty(arr.shape, arr.dtype)
        if arr.nbytes:
            if self.trace is not None:
                self.trace.append(("h2d", (arr.nbytes,)))
            _check(self.lib, self.lib.xh_memcpy_h2d(self.ctx, _vp(d.ptr), arr.ctypes.data_as(_vp), arr.nbytes))
        return d

    # ---- device-resident inputs across calls (the caller of the wrappers is the Indicator chain,
    # /root/reference/src/xclim/core/indicator.py:865-944: percentile_doy and the index that consumes its table read the
    # SAME field, and so do an index and the missing-value check after it) ----
    @staticmethod
    def _host_fingerprint(flat: np.ndarray):
        """Second guard against a buffer edited in place while it is remembered: the bit patterns of ~2^16 evenly spaced
        samples plus the first and last 4096 elements, summed as integers (every element of arrays up to 2^16 elements;
        ~0.5 ms on a 1.5 GB field against 27 ms for its upload).  An edit that touches none of the samples is NOT seen,
        and a full checksum costs as much as the transfer it would save (host memory bandwidth ~ PCIe bandwidth) — which
        is why the cache is SCOPED (:meth:`keep_inputs`) instead of trusting this."""
        n = flat.size
        step = max(1, n >> 16)
        u = flat.view(np.uint32 if flat.dtype.itemsize == 4 else np.uint64)
        return (n, int(u[::step].sum(dtype=np.uint64)), int(u[:4096].sum(dtype=np.uint64)), int(u[-4096:].sum(dtype=np.uint64)))

    def keep_inputs(self):
        """Context manager: inside it, large host inputs stay on the device from one call to the next
        (:meth:`resident`); at the exit of the outermost scope every copy is dropped.  The contract of a scope: no field
        handed to this library is edited IN PLACE while the scope is open (or :meth:`forget_inputs` is called after the
        edit).  ``patch.install()`` opens one around every ``Indicator.__call__`` — compute, its ``percentile_doy`` /
        ``resample_doy`` helpers and the missing-value check read the same DataArray and no user code runs in between;
        user code chains calls with ``with xclim_amd.keep_inputs(): per = percentile_doy(tasmax); out = tx90p(tasmax, per)``.
        Outside a scope every call uploads its inputs (the behaviour before round 5) unless XCLIM_AMD_INPUT_CACHE=always."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            with self.lock:
                self._keep_depth += 1
            try:
                yield self
            finally:
                with self.lock:
                    self._keep_depth -= 1
                    last = self._keep_depth == 0
                if last and self._inputs_mode != "always":
                    self.forget_inputs()

        return scope()

    def _inputs_active(self) -> bool:
        return self._inputs_mode == "always" or (self._inputs_mode == "scope" and self._keep_depth > 0)

    def resident(self, arr: np.ndarray) -> DeviceArray:
        """Device copy of a C-contiguous float32 / float64 host array.  Inside a :meth:`keep_inputs` scope large arrays
        (>= XCLIM_AMD_INPUT_CACHE_MIN bytes, default 32 MiB) are remembered by (address, shape, dtype) together with a weak
        reference to the object that owns the memory and a content fingerprint; the next call with the same buffer gets
        the SAME device memory (a view: the kernels only read their inputs) instead of another PCIe transfer.  The cache
        holds at most XCLIM_AMD_INPUT_CACHE_BYTES (default: half of the device memory free at first use; 0 = off), least
        recently used first out — also when an allocation fails (:meth:`empty`) —, and an entry dies with its host array
        or with the scope.  Outside a scope: a plain upload."""
        import weakref

        if (not self._inputs_active() or not isinstance(arr, np.ndarray) or not arr.flags.c_contiguous
                or arr.nbytes < self._inputs_min or arr.dtype not in (np.float32, np.float64)):
            return self.to_device(arr)
        if self._inputs_cap is None:
            self._inputs_cap = self.mem_info()[0] // 2
        if arr.nbytes > self._inputs_cap:
            return self.to_device(arr)
        owner = arr
        while isinstance(owner.base, np.ndarray):
            owner = owner.base
        if owner.base is not None:
            owner = owner.base  # (a buffer object under the outermost ndarray: pinned memory of pinned_empty)
        key = (arr.ctypes.data, arr.nbytes, arr.dtype.str)
        fp = self._host_fingerprint(arr.reshape(-1))
        with self.lock:
            hit = self._inputs.get(key)
            if hit is not None and hit[0]() is owner and hit[2] == fp:
                self._inputs[key] = self._inputs.pop(key)  # most recently used last
                if self.trace is not None:
                    self.trace.append(("resident_hit", (arr.nbytes,)))
                view = DeviceArray(self, hit[1].ptr, arr.shape, arr.dtype, owner=False)
                view._owner = hit[1]
                return view
            if hit is not None:
                self._drop_input(key)
        try:
            ref = weakref.ref(owner)
        except TypeError:
            return self.to_device(arr)
        d = self.to_device(arr)
        with self.lock:
            while self._inputs and self._inputs_bytes + arr.nbytes > self._inputs_cap:
                self._drop_input(next(iter(self._inputs)))
            self._inputs[key] = (ref, d, fp)
            self._inputs_bytes += arr.nbytes
        weakref.finalize(owner, self._drop_input, key, id(d))
        view = DeviceArray(self, d.ptr, arr.shape, arr.dtype, owner=False)
        view._owner = d
        return view

    def _drop_input(self, key, only_id=None) -> None:
        with self.lock:
            hit = self._inputs.get(key)
            if hit is None or (only_id is not None and id(hit[1]) != only_id):
                return
            del self._inputs[key]
            self._inputs_bytes -= hit[1].nbytes
        # (the device memory goes back to the pool when the last view of it is gone)

    def forget_inputs(self) -> None:
        """Drop every remembered input copy (after editing a field in place, or to give the memory back)."""
        with self.lock:
            self._inputs.clear()
            self._inputs_bytes = 0
        for hook in list(self._forget_hooks):
            hook()

    def wrap(self, ptr: int, shape, dtype) -> DeviceArray:
        """Wrap foreign device memory (e.g. a torch tensor's data_ptr()) without taking ownership."""
        return DeviceArray(self, int(ptr), shape, dtype, owner=False)

    def sync(self):
        _check(self.lib, self.lib.xh_sync(self.ctx))

    def name(self) -> str:
        buf = C.create_string_buffer(256)
        _check(self.lib, self.lib.xh_device_name(self.ctx, buf, 256))
        return buf.value.decode()

    def mem_info(self) -> tuple[int, int]:
        f, t = _sz(0), _sz(0)
        _check(self.lib, self.lib.xh_mem_info(self.ctx, C.byref(f), C.byref(t)))
        return f.value, t.value

    def timer_start(self):
        _check(self.lib, self.lib.xh_timer_start(self.ctx))

    def timer_stop(self) -> float:
        ms = _flt(0)
        _check(self.lib, self.lib.xh_timer_stop(self.ctx, C.byref(ms)))
        return float(ms.value)

    def close(self):
        if self.ctx:
            self.forget_inputs()
            self.trim()
            self.lib.xh_destroy(self.ctx)
            self.ctx = None

    def call(self, name: str, *args):
        """Invoke an entry point with this context as first argument and raise on error."""
        if self.trace is not None:  # launch log for the adapter tests: (entry point, arguments as passed)
            self.trace.append((name, args))
        with self.lock:
            _check(self.lib, getattr(self.lib, name)(self.ctx, *args))

    def start_trace(self) -> list:
        """Record every C-ABI call made through this context from now on: returns the (growing) list of
        ``(entry point, args)``; :meth:`stop_trace` ends it.  Test instrumentation (tests/test_gpu_adapter.py asserts that
        tx90p through the xarray wrappers reaches ``xh_threshold_count`` with the per-doy table, cdd ``xh_run_stats``)."""
        self.trace = []
        return self.trace

    def stop_trace(self) -> None:
        self.trace = None


_default_device: Device | None = None


def get_device(index: int | None = None) -> Device:
    """Process-wide default context (device from XCLIM_AMD_DEVICE / LOCAL_RANK, else 0)."""
    global _default_device
    if index is None:
        index = int(os.environ.get("XCLIM_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if _default_device is None or _default_device.index != index or _default_device.ctx is None:
        _default_device = Device(index)
    return _default_device


def np_ptr(a: np.ndarray) -> _vp:
    return a.ctypes.data_as(_vp)
