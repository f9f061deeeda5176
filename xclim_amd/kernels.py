"""Typed Python wrappers over the C ABI (one function per entry point of include/xclim_hip.h).

Inputs are :class:`~xclim_amd._capi.DeviceArray` views of shape (T, C) (time-major, C contiguous) unless stated;
numpy inputs are uploaded.  Outputs stay on the device (call ``.get()``), so chained ops (percentile_doy ->
threshold_count) never cross PCIe.  No CPU fallback exists here by design.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi as capi
from ._capi import OPS, REDUCERS, RUN_STATS, Device, DeviceArray, np_ptr

_vp = C.c_void_p


def as_device(dev: Device, a, dtype=np.float32) -> DeviceArray:
    if isinstance(a, DeviceArray):
        if a.dtype != np.dtype(dtype):
            raise TypeError(f"device array dtype {a.dtype} != {np.dtype(dtype)}")
        return a
    return dev.to_device(np.asarray(a), dtype=dtype)


def _tc(x: DeviceArray, dtype=np.float32):
    """Shape of a 2-D device field; the kernels read raw pointers, so the element type is checked here."""
    if len(x.shape) != 2:
        raise ValueError(f"expected a 2-D (time, cells) array, got shape {x.shape}")
    if dtype is not None and np.dtype(x.dtype) != np.dtype(dtype):
        raise TypeError(f"expected a {np.dtype(dtype).name} device array, got {np.dtype(x.dtype).name}")
    return int(x.shape[0]), int(x.shape[1])


def _seg(seg_off):
    s = np.ascontiguousarray(seg_off, dtype=np.int64)
    return s, len(s) - 1


def op_code(op: str) -> int:
    if op == "gteq":
        op = "ge"
    if op == "lteq":
        op = "le"
    if op not in OPS:
        raise ValueError(f"Operation `{op}` not recognized.")
    return OPS[op]


def fill_synthetic(dev: Device, T, C_, kind, seed, base, amp, p_wet=0.3, nan_per_million=0, cell0=0) -> DeviceArray:
    out = dev.empty((T, C_), np.float32)
    dbase = as_device(dev, np.asarray(base, dtype=np.float32))
    dev.call("xh_fill_synthetic", _vp(out.ptr), T, C_, C_, int(kind), int(seed), int(cell0), _vp(dbase.ptr), float(amp),
             float(p_wet), int(nan_per_million))
    dev.sync()
    return out


def transpose(dev: Device, x: DeviceArray) -> DeviceArray:
    r, c = _tc(x)
    out = dev.empty((c, r), np.float32)
    dev.call("xh_transpose_f32", _vp(x.ptr), r, c, c, _vp(out.ptr), r)
    return out


def threshold_count(dev: Device, x: DeviceArray, op: str, seg_off, *, scalar=None, scalar_f64=False, doy_table=None,
                    tidx=None, full=None, want_valid=True, out=None):
    """xh_threshold_count.  Exactly one of scalar / (doy_table, tidx) / full.  Returns (count, valid) (P, C) int32."""
    f64 = x.dtype == np.float64  # float64 field: float64 compare against any threshold (numpy promotion), xh_*_f64
    T, C_ = _tc(x, np.float64 if f64 else np.float32)
    seg, P = _seg(seg_off)
    if out is not None:
        count, valid = out
    else:
        count = dev.empty((P, C_), np.int32)
        valid = dev.empty((P, C_), np.int32) if want_valid else None
    table_ptr, tstride, tidx_ptr, kind, thr = _vp(0), 0, _vp(0), capi.THR_SCALAR_F32, 0.0
    keep = []
    if scalar is not None:
        kind = capi.THR_SCALAR_F64 if scalar_f64 else capi.THR_SCALAR_F32
        thr = float(scalar)
    elif doy_table is not None:
        kind = capi.THR_DOY_F64 if doy_table.dtype == np.float64 else capi.THR_DOY_F32
        table_ptr, tstride = _vp(doy_table.ptr), int(doy_table.shape[-1])
        if isinstance(tidx, DeviceArray):
            dt = tidx
        else:
            dt = dev.to_device(np.ascontiguousarray(tidx, dtype=np.int32))
            keep.append(dt)
        tidx_ptr = _vp(dt.ptr)
    elif full is not None:
        kind = capi.THR_FULL_F64 if full.dtype == np.float64 else capi.THR_FULL_F32
        table_ptr, tstride = _vp(full.ptr), int(full.shape[-1])
    else:
        raise ValueError("a threshold is required")
    if f64:
        if kind in (capi.THR_DOY_F32, capi.THR_FULL_F32):
            raise TypeError("a float64 field needs float64 threshold tables (a float32 table widens exactly: upload it as float64)")
        kind = capi.THR_SCALAR_F64 if kind == capi.THR_SCALAR_F32 else kind
    if kind == capi.THR_DOY_F64 and not f64:
        # the table's row count is known here: the multi-year tile kernel (tcount.hip) needs it to size its LDS slice
        dev.call("xh_threshold_count_doy", _vp(x.ptr), T, C_, C_, 1, op_code(op), table_ptr, tstride,
                 int(np.prod(doy_table.shape[:-1])), tidx_ptr, np_ptr(seg), P, _vp(count.ptr), _vp(valid.ptr if valid else 0))
    else:
        dev.call("xh_threshold_count_f64" if f64 else "xh_threshold_count", _vp(x.ptr), T, C_, C_, 1, op_code(op), kind, thr,
                 table_ptr, tstride, tidx_ptr, np_ptr(seg), P, _vp(count.ptr), _vp(valid.ptr if valid else 0))
    if keep:
        dev.sync()
    return count, valid


def domain_count(dev: Device, x: DeviceArray, op1, thr1, op2, thr2, combine, seg_off, want_valid=True):
    T, C_ = _tc(x)
    seg, P = _seg(seg_off)
    count = dev.empty((P, C_), np.int32)
    valid = dev.empty((P, C_), np.int32) if want_valid else None
    dev.call("xh_domain_count", _vp(x.ptr), T, C_, C_, 1, op_code(op1), float(thr1), op_code(op2), float(thr2),
             {"and": 1, "or": 2}[combine], np_ptr(seg), P, _vp(count.ptr), _vp(valid.ptr if valid else 0))
    return count, valid


def resample_reduce(dev: Device, x: DeviceArray, reducer: str, seg_off, skipna=True, want_valid=True):
    f64 = x.dtype == np.float64  # float64 field -> float64 statistics (xh_resample_reduce_f64)
    T, C_ = _tc(x, np.float64 if f64 else np.float32)
    seg, P = _seg(seg_off)
    if reducer not in REDUCERS:
        raise ValueError(f"Reducer `{reducer}` not recognized.")
    odt = np.int32 if reducer in ("count", "argmin", "argmax") else (np.float64 if f64 else np.float32)
    out = dev.empty((P, C_), odt)
    valid = dev.empty((P, C_), np.int32) if want_valid else None
    dev.call("xh_resample_reduce_f64" if f64 else "xh_resample_reduce", _vp(x.ptr), T, C_, C_, 1, REDUCERS[reducer], int(bool(skipna)), np_ptr(seg), P,
             _vp(out.ptr), _vp(valid.ptr if valid else 0))
    return out, valid


def apply_missing_mask(dev: Device, value: DeviceArray, valid: DeviceArray, expected, out=None) -> DeviceArray:
    P, C_ = _tc(value, None)
    exp = np.ascontiguousarray(expected, dtype=np.int32)
    assert exp.shape == (P,)
    out = out if out is not None else dev.empty((P, C_), np.float64)
    kind = 0 if value.dtype == np.int32 else 1
    dev.call("xh_apply_missing_mask", _vp(value.ptr), kind, _vp(valid.ptr), np_ptr(exp), P, C_, _vp(out.ptr))
    return out


def rolling_reduce(dev: Device, x: DeviceArray, window: int, reducer: str, center=True) -> DeviceArray:
    T, C_ = _tc(x)
    out = dev.empty((T, C_), np.float32)
    dev.call("xh_rolling_reduce", _vp(x.ptr), T, C_, C_, 1, int(window), int(bool(center)), REDUCERS[reducer],
             _vp(out.ptr), C_)
    return out


def rolling_dot(dev: Device, x: DeviceArray, weights) -> DeviceArray:
    """xh_rolling_dot: trailing weighted window sum (float64 sum, float32 result, NaN until the window is full)."""
    T, C_ = _tc(x)
    w = np.ascontiguousarray(weights, dtype=np.float64)
    out = dev.empty((T, C_), np.float32)
    dev.call("xh_rolling_dot", _vp(x.ptr), T, C_, C_, 1, len(w), np_ptr(w), _vp(out.ptr), C_)
    return out


def cumsum_reset(dev: Device, x: DeviceArray, index="last") -> DeviceArray:
    T, C_ = _tc(x)
    out = dev.empty((T, C_), np.float32)
    dev.call("xh_cumsum_reset", _vp(x.ptr), T, C_, C_, 1, int(index == "first"), _vp(out.ptr), C_)
    return out


def rle(dev: Device, x: DeviceArray, index="first") -> DeviceArray:
    T, C_ = _tc(x)
    out = dev.empty((T, C_), np.float32)
    dev.call("xh_rle", _vp(x.ptr), T, C_, C_, 1, int(index == "first"), _vp(out.ptr), C_)
    return out


def run_stats(dev: Device, x: DeviceArray, stat: str, window: int, seg_off, *, cut=True, index="first", fused_op=None,
              thresh=0.0, want_valid=True, out=None, one_dim=False):
    """xh_run_stats.  ``one_dim``: the NaN semantics of the reference's 1-D ufunc path (index "first" only); True = the
    windowed_run_count / windowed_run_events form, "stat" = the statistics_run_1d form (NaN for a series with NaN steps and
    no qualifying run)."""
    T, C_ = _tc(x)
    seg, P = _seg(seg_off)
    if out is not None:
        out, valid = out
    else:
        out = dev.empty((P, C_), np.float32)
        valid = dev.empty((P, C_), np.int32) if want_valid else None
    fop = -1 if fused_op is None else op_code(fused_op)
    dev.call("xh_run_stats", _vp(x.ptr), T, C_, C_, 1, fop, float(thresh), int(window), RUN_STATS[stat],
             ((3 if one_dim == "stat" else 2) if one_dim else 1) if index == "first" else 0, np_ptr(seg), P, int(bool(cut)), _vp(out.ptr),
             _vp(valid.ptr if valid else 0))
    return out, valid


def bivariate_count(dev: Device, x1: DeviceArray, x2: DeviceArray, op1, thr1, op2, thr2, combine, seg_off, want_valid=True):
    T, C_ = _tc(x1)
    assert x2.shape == x1.shape
    seg, P = _seg(seg_off)
    count = dev.empty((P, C_), np.int32)
    valid = dev.empty((P, C_), np.int32) if want_valid else None
    dev.call("xh_bivariate_count", _vp(x1.ptr), _vp(x2.ptr), T, C_, C_, C_, op_code(op1), float(thr1), op_code(op2), float(thr2),
             {"all": 1, "and": 1, "any": 2, "or": 2}[combine], np_ptr(seg), P, _vp(count.ptr), _vp(valid.ptr if valid else 0))
    return count, valid


def range_reduce(dev: Device, low: DeviceArray, high: DeviceArray, mode: str, reducer: str, seg_off, want_valid=True):
    """mode: "range" (reducer of high - low) | "interday" (mean |diff|) | "extreme" (max(high) - min(low))."""
    T, C_ = _tc(low)
    assert high.shape == low.shape
    seg, P = _seg(seg_off)
    out = dev.empty((P, C_), np.float32)
    valid = dev.empty((P, C_), np.int32) if want_valid else None
    dev.call("xh_range_reduce", _vp(low.ptr), _vp(high.ptr), T, C_, C_, C_, {"range": 0, "interday": 1, "extreme": 2}[mode],
             REDUCERS.get(reducer, 0), np_ptr(seg), P, _vp(out.ptr), _vp(valid.ptr if valid else 0))
    return out, valid


def compare_map(dev: Device, a: DeviceArray, op, thr, kind: str = "mask") -> DeviceArray:
    """kind: "mask" (uint8) | "maskf" (float 1/0) | "events" (float 1/0/NaN) | "where" (a where cond else NaN) | "excess"
    ((a - thr).clip(0), NaN kept; op unused); thr: scalar or DeviceArray."""
    T, C_ = _tc(a)
    ok = {"mask": 0, "events": 1, "where": 2, "maskf": 3, "excess": 4}[kind]
    out = dev.empty(a.shape, np.uint8 if ok == 0 else np.float32)
    if isinstance(thr, DeviceArray):
        assert thr.shape == a.shape and thr.dtype == np.float32
        dev.call("xh_compare_map", _vp(a.ptr), T, C_, C_, op_code(op), 0.0, 0, _vp(thr.ptr), C_, ok, _vp(out.ptr), C_)
    else:
        dev.call("xh_compare_map", _vp(a.ptr), T, C_, C_, op_code(op), float(thr), int(isinstance(thr, np.float64)), _vp(0), 0, ok,
                 _vp(out.ptr), C_)
    return out


def select_rows(dev: Device, x: DeviceArray, idx, out: DeviceArray | None = None, out_row: int = 0, out_stride_rows: int = 1) -> DeviceArray:
    """out row i = x row idx[i], NaN where idx[i] < 0 (xh_select_rows).  With ``out`` (rows, C): row i goes to row
    ``out_row + i * out_stride_rows`` of it (a strided scatter into an existing sample matrix)."""
    T, C_ = _tc(x)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    if out is None:
        out = dev.empty((len(idx), C_), np.float32)
        dev.call("xh_select_rows", _vp(x.ptr), T, C_, C_, 1, np_ptr(idx), len(idx), _vp(out.ptr), C_)
        return out
    rows, Co = _tc(out)
    if Co != C_ or out_row < 0 or (len(idx) and out_row + (len(idx) - 1) * out_stride_rows >= rows) or out_stride_rows < 1:
        raise ValueError("select_rows: the strided destination does not fit `out`")
    dev.call("xh_select_rows", _vp(x.ptr), T, C_, C_, 1, np_ptr(idx), len(idx), _vp(out.ptr + out_row * C_ * 4), out_stride_rows * C_)
    return out


def mask_to_f32(dev: Device, mask: DeviceArray) -> DeviceArray:
    """uint8 / bool device mask -> float32 1 / 0 mask of the same shape (xh_mask_u8_to_f32)."""
    if np.dtype(mask.dtype).itemsize != 1:
        raise TypeError(f"expected a 1-byte mask, got {np.dtype(mask.dtype).name}")
    out = dev.empty(mask.shape, np.float32)
    dev.call("xh_mask_u8_to_f32", _vp(mask.ptr), int(mask.size), _vp(out.ptr))
    return out


def thresholded_reduce(dev: Device, x: DeviceArray, op, thr, mode: int, reducer: str, seg_off, want_valid=True):
    T, C_ = _tc(x)
    seg, P = _seg(seg_off)
    out = dev.empty((P, C_), np.float32)
    valid = dev.empty((P, C_), np.int32) if want_valid else None
    dev.call("xh_thresholded_reduce", _vp(x.ptr), T, C_, C_, 1, op_code(op), float(thr), int(mode), REDUCERS.get(reducer, 0),
             np_ptr(seg), P, _vp(out.ptr), _vp(valid.ptr if valid else 0))
    return out, valid


def mask_rows(dev: Device, x: DeviceArray, seg_off, lo, hi, invert=False) -> DeviceArray:
    T, C_ = _tc(x)
    seg, P = _seg(seg_off)
    lo = np.ascontiguousarray(lo, dtype=np.int32)
    hi = np.ascontiguousarray(hi, dtype=np.int32)
    out = dev.empty((T, C_), np.float32)
    dev.call("xh_mask_rows", _vp(x.ptr), T, C_, C_, 1, np_ptr(seg), P, np_ptr(lo), np_ptr(hi), int(bool(invert)), _vp(out.ptr), C_)
    return out


def doy_mean_std(dev: Device, x: DeviceArray, tbase, window: int):
    T, C_ = _tc(x)
    tb = np.ascontiguousarray(tbase, dtype=np.int32)
    nyears, ndoy = tb.shape
    m, s = dev.empty((ndoy, C_), np.float32), dev.empty((ndoy, C_), np.float32)
    dev.call("xh_doy_mean_std", _vp(x.ptr), T, C_, C_, 1, np_ptr(tb), nyears, ndoy, int(window), _vp(m.ptr), _vp(s.ptr))
    return m, s


WIN_REDUCERS = {"sum": 0, "mean": 1, "min": 2, "max": 3, "wmean": 4}


def spell_mask(dev: Device, x: DeviceArray, window: int, win_reducer: str, op: str, thresh: float, weights=None) -> DeviceArray:
    T, C_ = _tc(x)
    out = dev.empty((T, C_), np.float32)
    w = np.ascontiguousarray(weights, dtype=np.float32) if weights is not None else None
    red = WIN_REDUCERS["wmean" if w is not None else (win_reducer or "min")]
    dev.call("xh_spell_mask", _vp(x.ptr), T, C_, C_, 1, int(window), red, op_code(op), float(thresh),
             np_ptr(w) if w is not None else _vp(0), _vp(out.ptr), C_)
    return out


def spell_run_stats(dev: Device, x: DeviceArray, window: int, win_reducer: str, op: str, thresh: float, stat: str, seg_off,
                    weights=None, want_valid=True):
    """xh_spell_run_stats: run statistics of the spell mask, the mask itself is never written (runs cut at the periods).
    Returns None when the shape is not covered (window > 8): the caller then takes spell_mask + run_stats."""
    from ._capi import XH_ERR_NOTIMPL, XclimHipError

    T, C_ = _tc(x)
    seg, P = _seg(seg_off)
    out = dev.empty((P, C_), np.float32)
    valid = dev.empty((P, C_), np.int32) if want_valid else None
    w = np.ascontiguousarray(weights, dtype=np.float32) if weights is not None else None
    red = WIN_REDUCERS["wmean" if w is not None else (win_reducer or "min")]
    try:
        dev.call("xh_spell_run_stats", _vp(x.ptr), T, C_, C_, 1, int(window), red, op_code(op), float(thresh),
                 np_ptr(w) if w is not None else _vp(0), RUN_STATS[stat], np_ptr(seg), P, _vp(out.ptr),
                 _vp(valid.ptr if valid else 0))
    except XclimHipError as e:
        if e.code == XH_ERR_NOTIMPL:
            return None
        raise
    return out, valid


def spell_mask_multi(dev: Device, xs, window: int, win_reducer: str, op: str, threshs, var_reducer="all", weights=None) -> DeviceArray:
    """spell_mask on a list of (T, C) device arrays with one threshold each (xh_spell_mask_multi)."""
    T, C_ = _tc(xs[0])
    assert all(x.shape == xs[0].shape for x in xs) and len(threshs) == len(xs)
    out = dev.empty((T, C_), np.float32)
    w = np.ascontiguousarray(weights, dtype=np.float32) if weights is not None else None
    red = WIN_REDUCERS["wmean" if w is not None else (win_reducer or "min")]
    ptrs = np.array([x.ptr for x in xs], dtype=np.uint64)
    thr = np.ascontiguousarray(threshs, dtype=np.float64)
    dev.call("xh_spell_mask_multi", np_ptr(ptrs), len(xs), np_ptr(thr), {"all": 1, "any": 2}[var_reducer], T, C_, C_, 1,
             int(window), red, op_code(op), np_ptr(w) if w is not None else _vp(0), _vp(out.ptr), C_)
    return out


def runs_with_holes(dev: Device, start: DeviceArray, window_start: int, stop: DeviceArray | None, window_stop: int) -> DeviceArray:
    T, C_ = _tc(start)
    out = dev.empty((T, C_), np.float32)
    dev.call("xh_runs_with_holes", _vp(start.ptr), _vp(stop.ptr if stop is not None else 0), T, C_, C_, 1, int(window_start),
             int(window_stop), _vp(out.ptr), C_)
    return out


def run_events(dev: Device, runs: DeviceArray, seg_off, maxev: int, eff: DeviceArray | None = None,
               data: DeviceArray | None = None, want=("start", "end", "len")):
    """Event compaction (xh_run_events): dict of (P, maxev, C) float32 arrays, NaN past the last run."""
    T, C_ = _tc(runs)
    seg, P = _seg(seg_off)
    out = {k: dev.empty((P, int(maxev), C_), np.float32) for k in want}
    if "start" not in out:
        out["start"] = dev.empty((P, int(maxev), C_), np.float32)
    ptr = lambda k: _vp(out[k].ptr if k in out else 0)
    dev.call("xh_run_events", _vp(runs.ptr), _vp(eff.ptr if eff is not None else 0), _vp(data.ptr if data is not None else 0), T, C_,
             C_, 1, np_ptr(seg), P, int(maxev), ptr("start"), ptr("end"), ptr("len"), ptr("eff"), ptr("sum"))
    return out


def suspicious_run(dev: Device, x: DeviceArray, window: int, op=None, thresh=None) -> DeviceArray:
    T, C_ = _tc(x)
    out = dev.empty((T, C_), np.uint8)
    dev.call("xh_suspicious_run", _vp(x.ptr), T, C_, C_, 1, int(window), -1 if thresh is None else op_code(op),
             0.0 if thresh is None else float(thresh), _vp(out.ptr), C_)
    return out


def keep_longest_run(dev: Device, x: DeviceArray, seg_off) -> DeviceArray:
    T, C_ = _tc(x)
    seg, P = _seg(seg_off)
    out = dev.empty((T, C_), np.float32)
    dev.call("xh_keep_longest_run", _vp(x.ptr), T, C_, C_, 1, np_ptr(seg), P, _vp(out.ptr), C_)
    return out


def season(dev: Device, x: DeviceArray, window: int, seg_off, mid_idx=None):
    T, C_ = _tc(x)
    seg, P = _seg(seg_off)
    s, e, ln = (dev.empty((P, C_), np.float32) for _ in range(3))
    mid = np.ascontiguousarray(mid_idx, dtype=np.int32) if mid_idx is not None else None
    dev.call("xh_season", _vp(x.ptr), T, C_, C_, 1, int(window), np_ptr(seg), np_ptr(mid) if mid is not None else _vp(0), P,
             _vp(s.ptr), _vp(e.ptr), _vp(ln.ptr))
    return s, e, ln


def max_run_sum(dev: Device, x: DeviceArray, window: int, seg_off, cut=True) -> DeviceArray:
    """xh_max_run_sum; cut=False: the reference's resample-after semantics (runs cross the period edges)."""
    T, C_ = _tc(x)
    seg, P = _seg(seg_off)
    out = dev.empty((P, C_), np.float32)
    dev.call("xh_max_run_sum", _vp(x.ptr), T, C_, C_, 1, int(window), np_ptr(seg), P, int(bool(cut)), _vp(out.ptr))
    return out


def nan_quantile(dev: Device, x: DeviceArray, q, alpha=1.0, beta=1.0, sample_axis=0) -> DeviceArray:
    """x: (N, C) if sample_axis == 0 else (C, N).  Returns (nq, C) float64."""
    q = np.ascontiguousarray(np.atleast_1d(q), dtype=np.float64)
    f64 = x.dtype == np.float64  # float64 samples: `diff` in float64 (utl:486), xh_nan_quantile_f64
    if sample_axis == 0:
        N, C_ = _tc(x, np.float64 if f64 else np.float32)
        sn, sc = C_, 1
    else:
        C_, N = _tc(x, np.float64 if f64 else np.float32)
        sn, sc = 1, N
    out = dev.empty((len(q), C_), np.float64)
    dev.call("xh_nan_quantile_f64" if f64 else "xh_nan_quantile", _vp(x.ptr), N, C_, sn, sc, np_ptr(q), len(q), float(alpha), float(beta), _vp(out.ptr))
    return out


def weighted_quantile(dev: Device, x: DeviceArray, weights, q) -> DeviceArray:
    """xh_weighted_quantile: x (N, C) member-major -> (nq, C) float64."""
    N, C_ = _tc(x)
    w = np.ascontiguousarray(weights, dtype=np.float64)
    q = np.ascontiguousarray(np.atleast_1d(q), dtype=np.float64)
    if w.shape != (N,):
        raise ValueError(f"weights must have one value per member ({N}), got shape {w.shape}")
    out = dev.empty((len(q), C_), np.float64)
    dev.call("xh_weighted_quantile", _vp(x.ptr), N, C_, C_, 1, np_ptr(w), np_ptr(q), len(q), _vp(out.ptr))
    return out


def percentile_doy(dev: Device, x: DeviceArray, tbase, window: int, per, alpha=1.0 / 3, beta=1.0 / 3,
                   out=None, vmap=None) -> DeviceArray:
    """Returns (nper, ndoy, C) float64 — percentile_doy before the 366-day adjustment.

    `vmap` (int32[Tv], optional): virtual-day -> physical-row table; `tbase` then indexes virtual days."""
    T, C_ = _tc(x)
    tb = np.ascontiguousarray(tbase, dtype=np.int32)
    nyears, ndoy = tb.shape
    per = np.ascontiguousarray(np.atleast_1d(per), dtype=np.float64)
    out = out if out is not None else dev.empty((len(per), ndoy, C_), np.float64)
    if vmap is None:
        dev.call("xh_percentile_doy", _vp(x.ptr), T, C_, C_, 1, np_ptr(tb), nyears, ndoy, int(window), np_ptr(per), len(per),
                 float(alpha), float(beta), _vp(out.ptr))
    else:
        vm = np.ascontiguousarray(vmap, dtype=np.int32)
        dev.call("xh_percentile_doy_mapped", _vp(x.ptr), T, C_, C_, 1, np_ptr(tb), nyears, ndoy, int(window), np_ptr(per),
                 len(per), float(alpha), float(beta), np_ptr(vm), len(vm), _vp(out.ptr))
    return out


def percentile_doy_count(dev: Device, x: DeviceArray, tbase, window: int, per: float, op: str, doy_period, P: int,
                         alpha=1.0 / 3, beta=1.0 / 3, want_valid=True, out=None):
    """Fused percentile_doy + threshold_count (xh_percentile_doy_count): (count, valid) int32 (P, C), or None when the
    shape is not covered by the fused kernel (the caller then runs the two-step chain)."""
    from ._capi import XH_ERR_NOTIMPL, XclimHipError

    T, C_ = _tc(x)
    tb = np.ascontiguousarray(tbase, dtype=np.int32)
    nyears, ndoy = tb.shape
    dp = np.ascontiguousarray(doy_period, dtype=np.int32).reshape(-1)
    assert dp.shape == (nyears * ndoy,)  # period of every (year, doy) day, < 0 where the day is absent
    if out is not None:
        cnt, val = out
    else:
        cnt = dev.empty((int(P), C_), np.int32)
        val = dev.empty((int(P), C_), np.int32) if want_valid else None
    try:
        dev.call("xh_percentile_doy_count", _vp(x.ptr), T, C_, C_, 1, np_ptr(tb), nyears, ndoy, int(window), float(per),
                 float(alpha), float(beta), op_code(op), np_ptr(dp), int(P), _vp(cnt.ptr), _vp(val.ptr if val else 0))
    except XclimHipError as e:
        if e.code == XH_ERR_NOTIMPL:
            return None
        raise
    return cnt, val


def doy_interp(dev: Device, table: DeviceArray, i0, i1, dxn, dxs, xsrc=None) -> DeviceArray:
    """table (D_in, C) float64 -> (D_out, C) float64 (xh_doy_interp).  `xsrc`: dayofyear coordinate of the source rows for
    the interpolate_na step (None: uniform)."""
    D_in, C_ = _tc(table, np.float64)
    xs = None if xsrc is None else np.ascontiguousarray(xsrc, dtype=np.float64)
    assert xs is None or len(xs) == D_in
    i0 = np.ascontiguousarray(i0, dtype=np.int32)
    i1 = np.ascontiguousarray(i1, dtype=np.int32)
    dxn = np.ascontiguousarray(dxn, dtype=np.float64)
    dxs = np.ascontiguousarray(dxs, dtype=np.float64)
    out = dev.empty((len(i0), C_), np.float64)
    dev.call("xh_doy_interp", _vp(table.ptr), D_in, C_, np_ptr(i0), np_ptr(i1), np_ptr(dxn), np_ptr(dxs), len(i0),
             _vp(out.ptr), np_ptr(xs) if xs is not None else _vp(0))
    return out


def doy_broadcast(dev: Device, table: DeviceArray, tidx) -> DeviceArray:
    """(D, C) float64 per-doy table -> (T, C) float64 field, row tidx[t] at step t (resample_doy)."""
    D, C_ = table.shape
    tidx = np.ascontiguousarray(tidx, dtype=np.int32)
    out = dev.empty((len(tidx), C_), np.float64)
    dev.call("xh_doy_broadcast", _vp(table.ptr), D, C_, np_ptr(tidx), len(tidx), _vp(out.ptr))
    return out


def within_bnds_doy(dev: Device, x: DeviceArray, low: DeviceArray, high: DeviceArray, tidx) -> DeviceArray:
    T, C_ = _tc(x)
    D = low.shape[0]
    tidx = np.ascontiguousarray(tidx, dtype=np.int32)
    assert len(tidx) == T and low.shape == high.shape == (D, C_)
    out = dev.empty((T, C_), np.uint8)
    dev.call("xh_within_bnds_doy", _vp(x.ptr), T, C_, C_, 1, _vp(low.ptr), _vp(high.ptr), D, np_ptr(tidx), _vp(out.ptr))
    return out


def compare_doy(dev: Device, x: DeviceArray, op: str, table: DeviceArray, tidx) -> DeviceArray:
    """float32 1/0 mask of x[t] op table[tidx[t]] (fp64 compare, (D, C) float64 per-doy table)."""
    T, C_ = _tc(x)
    D = table.shape[0]
    tidx = np.ascontiguousarray(tidx, dtype=np.int32)
    assert len(tidx) == T and table.shape == (D, C_)
    out = dev.empty((T, C_), np.float32)
    dev.call("xh_compare_doy", _vp(x.ptr), T, C_, C_, 1, op_code(op), _vp(table.ptr), D, np_ptr(tidx), _vp(out.ptr), C_)
    return out


def run_stats_doy(dev: Device, x: DeviceArray, op: str, table: DeviceArray, tidx, stat: str, window: int, seg_off, want_valid=True):
    """xh_run_stats_doy: run statistics (cut at the period edges) of x[t] op table[tidx[t]]; table (D, C) float64."""
    T, C_ = _tc(x)
    D = table.shape[0]
    tidx = np.ascontiguousarray(tidx, dtype=np.int32)
    assert len(tidx) == T and table.shape == (D, C_) and table.dtype == np.float64
    seg, P = _seg(seg_off)
    out = dev.empty((P, C_), np.float32)
    valid = dev.empty((P, C_), np.int32) if want_valid else None
    dev.call("xh_run_stats_doy", _vp(x.ptr), T, C_, C_, 1, op_code(op), _vp(table.ptr), D, np_ptr(tidx), int(window), RUN_STATS[stat],
             np_ptr(seg), P, _vp(out.ptr), _vp(valid.ptr if valid else 0))
    return out, valid


def precip_over_doy(dev: Device, x: DeviceArray, op: str, thr: float, table: DeviceArray, tidx, seg_off, want=("count",),
                    want_valid=True):
    """xh_precip_over_doy: (count | frac | both, valid) against max(table[tidx[t]], thr); table (D, C) float64."""
    T, C_ = _tc(x)
    D = table.shape[0]
    tidx = np.ascontiguousarray(tidx, dtype=np.int32)
    assert len(tidx) == T and table.shape == (D, C_) and table.dtype == np.float64
    seg, P = _seg(seg_off)
    cnt = dev.empty((P, C_), np.int32) if "count" in want else None
    frac = dev.empty((P, C_), np.float32) if "frac" in want else None
    valid = dev.empty((P, C_), np.int32) if want_valid else None
    dev.call("xh_precip_over_doy", _vp(x.ptr), T, C_, C_, 1, op_code(op), float(thr), _vp(table.ptr), D, np_ptr(tidx), np_ptr(seg),
             P, _vp(frac.ptr if frac else 0), _vp(cnt.ptr if cnt else 0), _vp(valid.ptr if valid else 0))
    return cnt, frac, valid


def quantile_series(dev: Device, x: DeviceArray, q, time_axis=0, out=None) -> DeviceArray:
    """Per-cell quantiles of the whole series: x (T, C) [time_axis 0] or (C, T) [time_axis 1] -> (nq, C) float32."""
    q = np.ascontiguousarray(np.atleast_1d(q), dtype=np.float64)
    if time_axis == 0:
        T, C_ = _tc(x)
        st, sc = C_, 1
    else:
        C_, T = _tc(x)
        st, sc = 1, T
    if out is None:
        out = dev.empty((len(q), C_), np.float32)
    dev.call("xh_quantile_series", _vp(x.ptr), T, C_, st, sc, np_ptr(q), len(q), _vp(out.ptr))
    return out


def eqm_train(dev: Device, ref: DeviceArray, hist: DeviceArray, q, kind="+", time_axis=0, out=None):
    q = np.ascontiguousarray(np.atleast_1d(q), dtype=np.float64)
    if time_axis == 0:
        T, C_ = _tc(ref)
        st, sc = C_, 1
    else:
        C_, T = _tc(ref)
        st, sc = 1, T
    if out is not None:
        af, hq = out
    else:
        af = dev.empty((len(q), C_), np.float32)
        hq = dev.empty((len(q), C_), np.float32)
    dev.call("xh_eqm_train", _vp(ref.ptr), _vp(hist.ptr), T, C_, st, sc, np_ptr(q), len(q), {"+": 0, "*": 1}[kind],
             _vp(af.ptr), _vp(hq.ptr))
    return af, hq


def eqm_train_window(dev: Device, ref: DeviceArray, hist: DeviceArray, rows0, enter, leave, q, kind="+", out=None, normalised=False):
    """xh_eqm_train_window: EQM training over a sliding row sample (day-of-year groups with a window).  rows0 (n0,): the time
    steps of the first group's sample; enter / leave (G - 1, per): the steps that enter / leave from one group to the next (-1 =
    none).  Returns (af, hist_q) as (G, nq, C) device arrays, or None when the kernel does not take the shape (the caller
    gathers every group's sample and calls :func:`eqm_train`).  ``normalised=True`` (xh_dqm_train_window, the training of a
    detrended quantile mapping): the quantiles of the samples divided by / shifted by their own means; returns (af, hist_q,
    scaling, mu_hist), the last two (G, C) float64 (``out`` then holds four arrays)."""
    from ._capi import XH_ERR_NOTIMPL, XclimHipError

    q = np.ascontiguousarray(np.atleast_1d(q), dtype=np.float64)
    T, C_ = _tc(ref)
    rows0 = np.ascontiguousarray(rows0, dtype=np.int32)
    enter = np.ascontiguousarray(enter, dtype=np.int32).reshape(-1, enter.shape[-1] if np.ndim(enter) == 2 else 1)
    leave = np.ascontiguousarray(leave, dtype=np.int32).reshape(enter.shape)
    G, per = enter.shape[0] + 1, enter.shape[1]
    if out is not None:
        af, hq = out[:2]
    else:
        af = dev.empty((G, len(q), C_), np.float32)
        hq = dev.empty((G, len(q), C_), np.float32)
    res, extra, name = (af, hq), (), "xh_eqm_train_window"
    if normalised:
        # xh_dqm_train_window: the samples normalised by their own means (dqm_train); + scaling, mu_hist (G, C) float64
        sc, muh = out[2:4] if out is not None else (dev.empty((G, C_), np.float64), dev.empty((G, C_), np.float64))
        res, extra, name = (af, hq, sc, muh), (_vp(sc.ptr), _vp(muh.ptr)), "xh_dqm_train_window"
    try:
        dev.call(name, _vp(ref.ptr), _vp(hist.ptr), T, C_, C_, np_ptr(rows0), len(rows0), np_ptr(enter), np_ptr(leave),
                 G, per, np_ptr(q), len(q), {"+": 0, "*": 1}[kind], _vp(af.ptr), _vp(hq.ptr), *extra)
    except XclimHipError as e:
        if e.code == XH_ERR_NOTIMPL:
            return None
        raise
    return res


def eqm_train_groups(dev: Device, ref: DeviceArray, hist: DeviceArray, rows, offs, q, kind="+", normalised=False):
    """xh_eqm_train_groups / xh_dqm_train_groups: the training of ALL (small) groups in one launch per field — rows: the row
    numbers group after group, offs (G + 1).  Returns (af, hist_q) (G, nq, C) — with ``normalised`` (dqm_train: the samples
    normalised by their group's mean) also (scaling, mu_hist) (G, C) float64 — or None when a group has more than 64 rows
    (gather each group and call :func:`eqm_train`)."""
    from ._capi import XH_ERR_NOTIMPL, XclimHipError

    q = np.ascontiguousarray(np.atleast_1d(q), dtype=np.float64)
    T, C_ = _tc(ref)
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    offs = np.ascontiguousarray(offs, dtype=np.int64)
    G = len(offs) - 1
    if int(np.diff(offs).max(initial=0)) > 64:
        return None
    af = dev.empty((G, len(q), C_), np.float32)
    hq = dev.empty((G, len(q), C_), np.float32)
    res, extra, name = (af, hq), (), "xh_eqm_train_groups"
    if normalised:
        sc, muh = dev.empty((G, C_), np.float64), dev.empty((G, C_), np.float64)
        res, extra, name = (af, hq, sc, muh), (_vp(sc.ptr), _vp(muh.ptr)), "xh_dqm_train_groups"
    try:
        dev.call(name, _vp(ref.ptr), _vp(hist.ptr), T, C_, C_, np_ptr(rows), np_ptr(offs), G, np_ptr(q), len(q), {"+": 0, "*": 1}[kind],
                 _vp(af.ptr), _vp(hq.ptr), *extra)
    except XclimHipError as e:
        if e.code == XH_ERR_NOTIMPL:
            return None
        raise
    return res


def eqm_adjust(dev: Device, sim: DeviceArray, af: DeviceArray, hist_q: DeviceArray, kind="+", interp="nearest",
               extrapolation="constant", out: DeviceArray | None = None) -> DeviceArray:
    T, C_ = _tc(sim)
    nq = int(af.shape[0])
    scen = out if out is not None else dev.empty((T, C_), np.float32)
    dev.call("xh_eqm_adjust", _vp(sim.ptr), T, C_, C_, 1, _vp(af.ptr), _vp(hist_q.ptr), nq, {"+": 0, "*": 1, "factor": 2}[kind],
             {"nearest": 0, "linear": 1, "cubic": 2}[interp], {"constant": 0, "nan": 1}[extrapolation], _vp(scen.ptr), C_)
    return scen


def mask_doy_cells(dev: Device, x: DeviceArray, doy, start: DeviceArray, end: DeviceArray) -> DeviceArray:
    """xh_mask_doy_cells: x where doy[t] is inside the cell's [start, end] (wrapping when start > end), NaN elsewhere."""
    T, C_ = _tc(x)
    d = np.ascontiguousarray(doy, dtype=np.int32)
    out = dev.empty((T, C_), np.float32)
    dev.call("xh_mask_doy_cells", _vp(x.ptr), T, C_, C_, 1, np_ptr(d), _vp(start.ptr), _vp(end.ptr), _vp(out.ptr), C_)
    return out


def mask_days_cells(dev: Device, x: DeviceArray, seg_off, lo: DeviceArray, hi: DeviceArray) -> DeviceArray:
    """xh_mask_days_cells: x where lo[p, c] <= t - seg_off[p] <= hi[p, c] (p = the period of step t), NaN elsewhere."""
    T, C_ = _tc(x)
    seg, P = _seg(seg_off)
    assert lo.shape == (P, C_) and hi.shape == (P, C_)
    out = dev.empty((T, C_), np.float32)
    dev.call("xh_mask_days_cells", _vp(x.ptr), T, C_, C_, 1, np_ptr(seg), P, _vp(lo.ptr), _vp(hi.ptr), _vp(out.ptr), C_)
    return out


def poly_trend(dev: Device, x: DeviceArray, degree: int = 1, u: DeviceArray | None = None):
    """xh_poly_trend: (p0, p1) float64 (C,) device arrays of the per-cell trend p0 + p1 (t - (T - 1) / 2); p1 None for degree 0.
    ``u`` (device float64, T): the rows' own coordinate instead of the centred row number (xh_poly_trend_u)."""
    T, C_ = _tc(x)
    p0 = dev.empty((C_,), np.float64)
    p1 = dev.empty((C_,), np.float64) if degree >= 1 else None
    if u is None:
        dev.call("xh_poly_trend", _vp(x.ptr), T, C_, C_, 1, int(degree), _vp(p0.ptr), _vp(p1.ptr if p1 else 0), _vp(0))
    else:
        dev.call("xh_poly_trend_u", _vp(x.ptr), T, C_, C_, 1, int(degree), _vp(u.ptr), _vp(p0.ptr), _vp(p1.ptr if p1 else 0), _vp(0))
    return p0, p1


def window_nanmean(dev: Device, x: DeviceArray, window: int, out: DeviceArray | None = None) -> DeviceArray:
    """xh_window_nanmean: the centred ``window``-step mean over the valid samples (the ends of the series see fewer rows)."""
    T, C_ = _tc(x)
    out = out if out is not None else dev.empty((T, C_), np.float32)
    dev.call("xh_window_nanmean", _vp(x.ptr), T, C_, C_, 1, int(window), _vp(out.ptr), C_)
    return out


def trend_apply(dev: Device, x: DeviceArray, p0: DeviceArray, p1, op: str, out: DeviceArray | None = None,
                u: DeviceArray | None = None) -> DeviceArray:
    """xh_trend_apply: x OP (p0[c] + p1[c] (t - (T - 1) / 2)), op in "+", "-", "*", "/"; p1 None: per-cell constant.
    ``u``: the rows' own coordinate (xh_trend_apply_u)."""
    T, C_ = _tc(x)
    out = out if out is not None else dev.empty((T, C_), np.float32)
    mode = {"+": 0, "-": 1, "*": 2, "/": 3}[op]
    if u is None:
        dev.call("xh_trend_apply", _vp(x.ptr), T, C_, C_, 1, _vp(p0.ptr), _vp(p1.ptr if p1 is not None else 0), mode, _vp(out.ptr), C_)
    else:
        dev.call("xh_trend_apply_u", _vp(x.ptr), T, C_, C_, 1, _vp(u.ptr), _vp(p0.ptr), _vp(p1.ptr if p1 is not None else 0), mode,
                 _vp(out.ptr), C_)
    return out


def qdm_adjust_groups(dev: Device, sim: DeviceArray, rows, offs, af_all: DeviceArray, q, kind="+", interp="nearest",
                      extrapolation="constant", out: DeviceArray | None = None):
    """xh_qdm_adjust_groups: QDM adjust of every (small) group of rows in one launch — rows: the row numbers group after group,
    offs (G + 1), af_all (G, nq, C).  Returns scen (T, C) (rows in no group keep what ``out`` held), or None when the kernel does
    not take the shape (a group of more than 64 rows, more than 32 nodes): gather each group and call :func:`qdm_adjust`."""
    from ._capi import XH_ERR_NOTIMPL, XclimHipError

    q = np.ascontiguousarray(np.atleast_1d(q), dtype=np.float64)
    if interp not in ("nearest", "linear"):
        return None
    T, C_ = _tc(sim)
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    offs = np.ascontiguousarray(offs, dtype=np.int64)
    if int(np.diff(offs).max(initial=0)) > 64 or len(q) > 32:
        return None
    scen = out if out is not None else dev.empty((T, C_), np.float32)
    try:
        dev.call("xh_qdm_adjust_groups", _vp(sim.ptr), T, C_, C_, np_ptr(rows), np_ptr(offs), len(offs) - 1, _vp(af_all.ptr), np_ptr(q), len(q),
                 {"+": 0, "*": 1, "factor": 2}[kind], {"nearest": 0, "linear": 1}[interp], {"constant": 0, "nan": 1}[extrapolation],
                 _vp(scen.ptr), C_)
    except XclimHipError as e:
        if e.code == XH_ERR_NOTIMPL:
            return None
        raise
    return scen


def poly_trend_groups(dev: Device, x: DeviceArray, rows, offs, u: DeviceArray, degree: int = 1):
    """xh_poly_trend_groups: the per-cell trend of every GROUP of rows in one launch.  rows: the row numbers group after group,
    offs (G + 1): where each group's rows start, u (device float64, T): the coordinate of every row.  (p0, p1): (G, C) float64
    device arrays (p1 None for degree 0)."""
    T, C_ = _tc(x)
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    offs = np.ascontiguousarray(offs, dtype=np.int64)
    G = len(offs) - 1
    p0 = dev.empty((G, C_), np.float64)
    p1 = dev.empty((G, C_), np.float64) if degree >= 1 else None
    dev.call("xh_poly_trend_groups", _vp(x.ptr), T, C_, C_, np_ptr(rows), np_ptr(offs), G, _vp(u.ptr), int(degree), _vp(p0.ptr),
             _vp(p1.ptr if p1 is not None else 0))
    return p0, p1


def trend_apply_groups(dev: Device, x: DeviceArray, rows, offs, p0: DeviceArray, p1, op: str, u: DeviceArray | None = None,
                       out: DeviceArray | None = None) -> DeviceArray:
    """xh_trend_apply_groups: x OP (p0[g, c] + p1[g, c] u[t]) for the rows t of every group g (p0, p1: (G, C) float64 device arrays;
    p1 None: a per-group constant).  Rows in no group keep what ``out`` held (a fresh ``out`` is NaN-free only if every row is
    listed)."""
    T, C_ = _tc(x)
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    offs = np.ascontiguousarray(offs, dtype=np.int64)
    out = out if out is not None else dev.empty((T, C_), np.float32)
    dev.call("xh_trend_apply_groups", _vp(x.ptr), T, C_, C_, np_ptr(rows), np_ptr(offs), len(offs) - 1, _vp(u.ptr if u is not None else 0),
             _vp(p0.ptr), _vp(p1.ptr if p1 is not None else 0), {"+": 0, "-": 1, "*": 2, "/": 3}[op], _vp(out.ptr), C_)
    return out


def qdm_adjust(dev: Device, sim: DeviceArray, af: DeviceArray, q, kind="+", interp="nearest", extrapolation="constant",
               time_axis=0, out: DeviceArray | None = None) -> DeviceArray:
    """xh_qdm_adjust: sim (T, C) [time_axis 0] or (C, T) [time_axis 1], af (nq, C), q the nq quantile nodes; scen in the
    layout of sim."""
    q = np.ascontiguousarray(np.atleast_1d(q), dtype=np.float64)
    if interp not in ("nearest", "linear"):
        raise NotImplementedError(f"qdm_adjust: interp={interp!r} (nearest and linear are built)")
    if time_axis == 0:
        T, C_ = _tc(sim)
        st, sc = C_, 1
    else:
        C_, T = _tc(sim)
        st, sc = 1, T
    if int(af.shape[0]) != len(q):
        raise ValueError("qdm_adjust: af must hold one row per quantile node")
    if np.any(np.diff(q) <= 0):
        raise ValueError("qdm_adjust: the quantile nodes must be strictly increasing")
    scen = out if out is not None else dev.empty(tuple(sim.shape), np.float32)
    dev.call("xh_qdm_adjust", _vp(sim.ptr), T, C_, st, sc, _vp(af.ptr), np_ptr(q), len(q), {"+": 0, "*": 1, "factor": 2}[kind],
             {"nearest": 0, "linear": 1}[interp], {"constant": 0, "nan": 1}[extrapolation], _vp(scen.ptr))
    return scen


def quantile_cells(dev: Device, x: DeviceArray, q_cell, time_axis=0) -> DeviceArray:
    """xh_quantile_cells (xsdba.nbutils.vecquantiles): one quantile per cell at its own probability `q_cell` (C float64,
    host or device); x (T, C) [time_axis 0] or (C, T)."""
    if time_axis == 0:
        T, C_ = _tc(x)
        st, sc = C_, 1
    else:
        C_, T = _tc(x)
        st, sc = 1, T
    qd = q_cell if isinstance(q_cell, DeviceArray) else dev.to_device(np.ascontiguousarray(q_cell, dtype=np.float64))
    out = dev.empty((C_,), np.float32)
    dev.call("xh_quantile_cells", _vp(x.ptr), T, C_, st, sc, _vp(qd.ptr), _vp(out.ptr))
    return out


def adapt_freq(dev: Device, sim: DeviceArray, p0_ref, p0_sim, dp0, pth, thresh: float, seed: int = 0, tindex=None, cell0: int = 0,
               out: DeviceArray | None = None) -> DeviceArray:
    """xh_adapt_freq: the value-replacement step of xsdba.processing.adapt_freq on sim (T, C); per-cell P0_ref / P0_sim /
    dP0 (float64) and pth (float32), host or device arrays of C."""
    T, C_ = _tc(sim)

    def up(a, dt):
        return a if isinstance(a, DeviceArray) else dev.to_device(np.ascontiguousarray(a, dtype=dt))

    pr, ps, dp, pt = up(p0_ref, np.float64), up(p0_sim, np.float64), up(dp0, np.float64), up(pth, np.float32)
    ti = None if tindex is None else dev.to_device(np.ascontiguousarray(tindex, dtype=np.int64))
    scen = out if out is not None else dev.empty(tuple(sim.shape), np.float32)
    dev.call("xh_adapt_freq", _vp(sim.ptr), T, C_, C_, 1, _vp(pr.ptr), _vp(ps.ptr), _vp(dp.ptr), _vp(pt.ptr), float(thresh),
             int(seed) & 0xFFFFFFFFFFFFFFFF, _vp(ti.ptr) if ti is not None else None, int(cell0), _vp(scen.ptr))
    return scen


def eqm_adjust_g2d(dev: Device, sim: DeviceArray, af_all: DeviceArray, hq_all: DeviceArray, gcoord: int, kind="+",
                   extrapolation="constant", out: DeviceArray | None = None) -> DeviceArray:
    """xh_eqm_adjust_g2d: the rows of ONE group (coordinate `gcoord` in 1 .. G) adjusted with xsdba's 2-D "nearest" over the
    nodes of all groups; af_all / hq_all (G, nq, C)."""
    n, C_ = _tc(sim)
    G, nq = int(af_all.shape[0]), int(af_all.shape[1])
    scen = out if out is not None else dev.empty((n, C_), np.float32)
    dev.call("xh_eqm_adjust_g2d", _vp(sim.ptr), n, C_, C_, _vp(af_all.ptr), _vp(hq_all.ptr), G, nq, int(gcoord),
             {"+": 0, "*": 1, "factor": 2}[kind], {"constant": 0, "nan": 1}[extrapolation], _vp(scen.ptr), C_)
    return scen


def plane_linear(dev: Device, xnew: DeviceArray, gnew, yq_all: DeviceArray, *, xq_all: DeviceArray | None = None, xq_common=None,
                 base: DeviceArray | None = None, kind="+", out: DeviceArray | None = None) -> DeviceArray:
    """xh_plane_linear: xsdba's 2-D ``interp_on_quantiles(method="linear")`` (Delaunay interpolation over the nodes of all
    groups, constant extrapolation).  xnew (T, C) abscissa of every step, gnew (T) float64 group coordinate (host or
    device), yq_all (G, nq, C) factors, node abscissae ``xq_all`` (G, nq, C) or ``xq_common`` (nq, host: QDM's quantiles);
    ``base`` (T, C): the values the factor is applied to (default xnew); kind "+" | "*" | "factor"."""
    T, C_ = _tc(xnew)
    G, nq = int(yq_all.shape[0]), int(yq_all.shape[1])
    if (xq_all is None) == (xq_common is None):
        raise ValueError("plane_linear: give xq_all or xq_common")
    gd = gnew if isinstance(gnew, DeviceArray) else dev.to_device(np.ascontiguousarray(gnew, dtype=np.float64))
    if int(gd.shape[0]) != T:
        raise ValueError("plane_linear: one group coordinate per time step")
    qc = None if xq_common is None else np.ascontiguousarray(xq_common, dtype=np.float64)
    if qc is not None and len(qc) != nq:
        raise ValueError("plane_linear: xq_common must hold one abscissa per node")
    scen = out if out is not None else dev.empty((T, C_), np.float32)
    dev.call("xh_plane_linear", _vp(xnew.ptr), _vp(base.ptr) if base is not None else None, T, C_, C_, _vp(gd.ptr),
             _vp(xq_all.ptr) if xq_all is not None else None, np_ptr(qc) if qc is not None else None, _vp(yq_all.ptr), G, nq,
             {"+": 0, "*": 1, "factor": 2}[kind], _vp(scen.ptr), C_)
    return scen


def plane_nearest(dev: Device, xnew: DeviceArray, gnew, yq_all: DeviceArray, xq_all: DeviceArray, kind="+", extrapolation="constant",
                  out: DeviceArray | None = None) -> DeviceArray:
    """xh_plane_nearest: xsdba's 2-D ``interp_on_quantiles(method="nearest")`` for a month / day-of-year grouping over the
    whole series in one call; gnew (T): the INTEGER group coordinate 1 .. G of every step; yq_all / xq_all (G, nq <= 32, C)."""
    T, C_ = _tc(xnew)
    G, nq = int(yq_all.shape[0]), int(yq_all.shape[1])
    gd = gnew if isinstance(gnew, DeviceArray) else dev.to_device(np.ascontiguousarray(gnew, dtype=np.float64))
    if int(gd.shape[0]) != T:
        raise ValueError("plane_nearest: one group coordinate per time step")
    scen = out if out is not None else dev.empty((T, C_), np.float32)
    dev.call("xh_plane_nearest", _vp(xnew.ptr), None, T, C_, C_, _vp(gd.ptr), _vp(xq_all.ptr), None, _vp(yq_all.ptr), G, nq,
             {"+": 0, "*": 1, "factor": 2}[kind], {"constant": 0, "nan": 1}[extrapolation], _vp(scen.ptr), C_)
    return scen


def apply_factor(dev: Device, base: DeviceArray, fac: DeviceArray, kind="+", out: DeviceArray | None = None) -> DeviceArray:
    """xh_apply_factor: base (+|*) fac, two (T, C) float32 fields."""
    T, C_ = _tc(base)
    out = out if out is not None else dev.empty((T, C_), np.float32)
    dev.call("xh_apply_factor", _vp(base.ptr), _vp(fac.ptr), T, C_, C_, C_, {"+": 0, "*": 1}[kind], _vp(out.ptr), C_)
    return out
