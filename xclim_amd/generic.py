"""Host mirror of ``xclim.indices.generic`` (reference: src/xclim/indices/generic.py) over the HIP kernels.

Same function names, argument meaning and error behaviour as the reference; differences forced by the missing
xarray: data are numpy/device arrays with TIME ON AXIS 0 and the time coordinate is a
:class:`~xclim_amd.timeaxis.TimeAxis`; thresholds are plain numbers in the units of the data (unit conversion is
pint/host work, out of scope).  Every function returns numpy (``(P, *cells)``) unless ``keep=True`` (device array
``(P, C)``), and can also return the fused MissingAny valid count (``with_valid=True``).
"""

from __future__ import annotations

import numpy as np

from . import kernels as K
from ._capi import handle_float64, OPS, DeviceArray, get_device
from .calendar import DoyPercentile, _flatten, adjust_doy_calendar, resample_doy_index
from .timeaxis import TimeAxis

binary_ops = {">": "gt", "<": "lt", ">=": "ge", "<=": "le", "==": "eq", "!=": "ne"}


def get_op(op: str, constrain=None) -> str:
    """gen:255-298: validate an operator name; returns the canonical symbol."""
    if op == "gteq":
        op = "ge"
    if op == "lteq":
        op = "le"
    if op in binary_ops:
        sym = op
    elif op in binary_ops.values():
        sym = {v: k for k, v in binary_ops.items()}[op]
    else:
        raise ValueError(f"Operation `{op}` not recognized.")
    if constrain:
        if isinstance(constrain, str):  # gen:289-290 accepts a bare string: do not iterate over its characters
            constrain = [constrain]
        allowed = list(constrain) + [binary_ops[c] for c in constrain if c in binary_ops]
        if op not in allowed:
            raise ValueError(f"Operation `{op}` not permitted for indice.")
    return sym


def _finish(out: DeviceArray, valid, cell_shape, keep, with_valid):
    if keep:
        return (out, valid) if with_valid else out
    o = out.get().reshape((out.shape[0],) + tuple(cell_shape))
    if with_valid:
        return o, valid.get().reshape(o.shape)
    return o


def _cell_threshold(dev, threshold, da):
    """A threshold with one value per grid cell (e.g. a percentile over time, a climatological field) as the one-row
    float64 table + all-zero row index that the per-doy kernels take — nothing of shape (T, cells) is materialised, and
    the float64 compare of exactly converted float32 values equals numpy's float32 compare.  None when `threshold` is
    not of that shape."""
    if isinstance(threshold, (DeviceArray, DoyPercentile)) or np.ndim(threshold) == 0:
        return None
    th = np.asarray(threshold)
    shape = tuple(da.shape) if isinstance(da, DeviceArray) else np.shape(da)
    if th.ndim == len(shape) and th.shape[0] == 1:
        th = th[0]
    if th.ndim == 0:
        return None
    if th.shape != tuple(shape[1:]):
        # a flattened device view (T, C) of the data: the threshold still has the caller's cell shape
        if not (isinstance(da, DeviceArray) and len(shape) == 2 and th.size == shape[1]):
            return None
    table = dev.to_device(np.ascontiguousarray(th, dtype=np.float64).reshape(1, -1))
    return table, np.zeros(shape[0], dtype=np.int32)


def threshold_count(da, op: str, threshold, time: TimeAxis, freq: str, constrain=None, *, device=None, keep=False,
                    with_valid=False):
    """gen:329-361.  ``threshold``: python float (fp32 compare, NumPy weak-scalar rule), ``np.float64`` scalar (fp64
    compare), a :class:`DoyPercentile` (resample_doy fused: per-doy fp64 table, cal:763-790) or a full (T, *cells)
    array / device array."""
    if constrain is None:
        constrain = (">", "<", ">=", "<=")
    sym = get_op(op, constrain)
    dev = device or get_device()
    x, cell_shape = _flatten(da, dev, f64=True)  # a float64 field is compared in float64 (xh_threshold_count_f64)
    seg, _ = time.segments(freq)
    if x.dtype == np.float64 and not isinstance(threshold, (DoyPercentile, DeviceArray)) and np.ndim(threshold) > 0 \
            and _cell_threshold(dev, threshold, da) is None:
        full = dev.to_device(np.broadcast_to(np.asarray(threshold), np.shape(da)).reshape(x.shape), dtype=np.float64)
        cnt, val = K.threshold_count(dev, x, sym, seg, full=full)
        return _finish(cnt, val, cell_shape, keep, with_valid)
    if isinstance(threshold, DoyPercentile):
        doy = adjust_doy_calendar(threshold, time, dev)
        if doy.data.shape[0] != 1:
            raise ValueError("select one percentile first (DoyPercentile.sel)")
        table = doy.data.reshape(doy.data.shape[1], doy.data.shape[2])
        cnt, val = K.threshold_count(dev, x, sym, seg, doy_table=table, tidx=resample_doy_index(doy, time))
    elif isinstance(threshold, DeviceArray):
        cnt, val = K.threshold_count(dev, x, sym, seg, full=threshold.reshape(threshold.shape[0], -1))
    elif _cell_threshold(dev, threshold, da) is not None:  # one value per cell
        table, tidx = _cell_threshold(dev, threshold, da)
        cnt, val = K.threshold_count(dev, x, sym, seg, doy_table=table, tidx=tidx)
    elif np.ndim(threshold) == 0:
        # NumPy 2 promotion: python float is a weak scalar (fp32 compare); an np.float64 scalar forces fp64
        f64 = isinstance(threshold, np.float64)
        cnt, val = K.threshold_count(dev, x, sym, seg, scalar=float(threshold), scalar_f64=bool(f64))
    else:
        th = np.asarray(threshold)
        dt = np.float64 if th.dtype == np.float64 else np.float32
        full = dev.to_device(np.broadcast_to(th, np.shape(da)).reshape(x.shape), dtype=dt)
        cnt, val = K.threshold_count(dev, x, sym, seg, full=full)
    return _finish(cnt, val, cell_shape, keep, with_valid)


def count_occurrences(da, threshold: float, op: str, time: TimeAxis, freq: str, constrain=None, **kw):
    """gen:960-999 (all six operators allowed)."""
    return threshold_count(da, op, threshold, time, freq, constrain=constrain or tuple(OPS), **kw)


def domain_count(da, low: float, high: float, time: TimeAxis, freq: str, *, device=None, keep=False, with_valid=False):
    """gen:364-392: count of ``low < da <= high`` per period."""
    dev = device or get_device()
    x, cell_shape = _flatten(da, dev)
    seg, _ = time.segments(freq)
    cnt, val = K.domain_count(dev, x, ">", low, "<=", high, "and", seg)
    return _finish(cnt, val, cell_shape, keep, with_valid)


def count_level_crossings(low_data, high_data, threshold: float, time: TimeAxis, freq: str, *, op_low: str = "<",
                          op_high: str = ">=", device=None, keep=False, with_valid=False):
    """gen:913-957: days with `low_data op_low threshold` AND `high_data op_high threshold` (freeze-thaw cycles)."""
    dev = device or get_device()
    lo, cell_shape = _flatten(low_data, dev)
    hi, _ = _flatten(high_data, dev)
    seg, _ = time.segments(freq)
    cnt, val = K.bivariate_count(dev, lo, hi, get_op(op_low, ("<", "<=")), threshold, get_op(op_high, (">", ">=")), threshold,
                                 "all", seg)
    return _finish(cnt, val, cell_shape, keep, with_valid)


def bivariate_count_occurrences(*, data_var1, data_var2, threshold_var1: float, threshold_var2: float, time: TimeAxis,
                                freq: str, op_var1: str, op_var2: str, var_reducer: str, constrain_var1=None,
                                constrain_var2=None, device=None, keep=False, with_valid=False):
    """gen:1002-1073: count days where the two conditions hold for `all` / `any` of the variables."""
    if var_reducer not in ("all", "any"):
        raise ValueError(f"Unsupported value for var_reducer: {var_reducer}")
    dev = device or get_device()
    a, cell_shape = _flatten(data_var1, dev)
    b, _ = _flatten(data_var2, dev)
    seg, _ = time.segments(freq)
    cnt, val = K.bivariate_count(dev, a, b, get_op(op_var1, constrain_var1), threshold_var1, get_op(op_var2, constrain_var2),
                                 threshold_var2, var_reducer, seg)
    return _finish(cnt, val, cell_shape, keep, with_valid)


def compare(left, op: str, right, constrain=None, *, device=None, keep=False):
    """gen:301-326: elementwise ``left op right`` -> bool mask (NaN compares False, True for ``!=``).  ``right``: python
    float (fp32 compare), ``np.float64`` (fp64 compare) or an array of the shape of ``left``."""
    sym = get_op(op, constrain)
    dev = device or get_device()
    x, cell_shape = _flatten(left, dev)
    # keep=True: a float32 1 / 0 mask, the form every device consumer (run_length.*, spell kernels) reads; otherwise a
    # uint8 mask (a quarter of the bytes over PCIe) turned into numpy bool
    kind = "maskf" if keep else "mask"
    if np.ndim(right) == 0 and not isinstance(right, DeviceArray):
        m = K.compare_map(dev, x, sym, right, kind)
    else:
        handle_float64(np.asarray(right), "compare: array threshold")
        b, _ = _flatten(np.broadcast_to(np.asarray(right, dtype=np.float32), np.shape(left))
                        if not isinstance(right, DeviceArray) else right, dev)
        m = K.compare_map(dev, x, sym, b, kind)
    if keep:
        return m
    return m.get().reshape((x.shape[0],) + tuple(cell_shape)).astype(bool)


def get_daily_events(da, threshold, op: str, constrain=None, *, device=None, keep=False):
    """gen:395-431: 1 where ``da op threshold``, 0 where not, NaN where ``da`` is NaN (float32)."""
    sym = get_op(op, constrain)
    dev = device or get_device()
    x, cell_shape = _flatten(da, dev)
    if np.ndim(threshold) == 0 and not isinstance(threshold, DeviceArray):
        ev = K.compare_map(dev, x, sym, threshold, "events")
    else:
        handle_float64(np.asarray(threshold), "compare: array threshold")
        b, _ = _flatten(np.broadcast_to(np.asarray(threshold, dtype=np.float32), np.shape(da))
                        if not isinstance(threshold, DeviceArray) else threshold, dev)
        ev = K.compare_map(dev, x, sym, b, "events")
    if keep:
        return ev
    return ev.get().reshape((x.shape[0],) + tuple(cell_shape))


def _range(low_data, high_data, mode, reducer, time, freq, device, keep, with_valid):
    dev = device or get_device()
    lo, cell_shape = _flatten(low_data, dev)
    hi, _ = _flatten(high_data, dev)
    seg, _ = time.segments(freq)
    out, val = K.range_reduce(dev, lo, hi, mode, reducer, seg)
    return _finish(out, val, cell_shape, keep, with_valid)


def diurnal_temperature_range(low_data, high_data, reducer: str, time: TimeAxis, freq: str, *, device=None, keep=False,
                              with_valid=False):
    """gen:1076-1105: ``reducer`` in {max, min, mean, sum} of (high - low) per period."""
    if reducer not in ("max", "min", "mean", "sum"):
        raise ValueError(f"Reducer `{reducer}` not supported.")
    return _range(low_data, high_data, "range", reducer, time, freq, device, keep, with_valid)


def interday_diurnal_temperature_range(low_data, high_data, time: TimeAxis, freq: str, *, device=None, keep=False,
                                       with_valid=False):
    """gen:1360-1385: mean absolute day-to-day difference of the diurnal range."""
    return _range(low_data, high_data, "interday", "mean", time, freq, device, keep, with_valid)


def extreme_temperature_range(low_data, high_data, time: TimeAxis, freq: str, *, device=None, keep=False,
                              with_valid=False):
    """gen:1388-1414: max of the daily maxima minus min of the daily minima."""
    return _range(low_data, high_data, "extreme", "max", time, freq, device, keep, with_valid)


def _thresholded(data, op, threshold, mode, reducer, time, freq, constrain, device, keep, with_valid):
    sym = get_op(op, constrain)
    dev = device or get_device()
    x, cell_shape = _flatten(data, dev)
    seg, _ = time.segments(freq)
    out, val = K.thresholded_reduce(dev, x, sym, float(threshold), mode, reducer, seg)
    return _finish(out, val, cell_shape, keep, with_valid)


def statistics(data, reducer: str, time: TimeAxis, freq: str, **kw):
    """gen:1255-1275: data.resample(time=freq).<max|min|mean|sum>()."""
    return select_resample_op(data, reducer, time, freq, **kw)


def thresholded_statistics(data, op: str, threshold: float, reducer: str, time: TimeAxis, freq: str, constrain=None, *,
                           device=None, keep=False, with_valid=False):
    """gen:1278-1320: reducer of the values that satisfy `data op threshold`."""
    if reducer not in ("max", "min", "mean", "sum"):
        raise ValueError(f"Reducer `{reducer}` not recognized.")
    return _thresholded(data, op, threshold, 0, reducer, time, freq, constrain, device, keep, with_valid)


def temperature_sum(data, op: str, threshold: float, time: TimeAxis, freq: str, *, device=None, keep=False, with_valid=False):
    """gen:1323-1357: direction * sum of (data - threshold) over the days satisfying the condition (degree-days)."""
    return _thresholded(data, op, threshold, 1, "sum", time, freq, ("<", "<=", ">", ">="), device, keep, with_valid)


def cumulative_difference(data, threshold: float, op: str, time: TimeAxis, freq: str, *, device=None, keep=False,
                          with_valid=False):
    """gen:1514-1552: sum over the period of the positive part of (data - threshold) or (threshold - data)."""
    if get_op(op) not in ("<", "<=", ">", ">="):
        raise NotImplementedError(f"Condition not supported: '{op}'.")
    return _thresholded(data, op, threshold, 2, "sum", time, freq, None, device, keep, with_valid)


def select_resample_op(da, op: str, time: TimeAxis, freq: str = "YS", *, device=None, keep=False, with_valid=False,
                       **indexer):
    """gen:83-125 (string ops): min/max/mean/std/var/count/sum/integral/argmax/argmin per period; ``**indexer``
    (season= / month= / doy_bounds= / date_bounds=) masks the other time steps first (calendar.select_time)."""
    dev = device or get_device()
    x, cell_shape = _flatten(da, dev, f64=not indexer)  # float64 field -> float64 statistics (xh_resample_reduce_f64)
    if indexer:
        from .calendar import select_time

        x = select_time(x, time, device=dev, keep=True, **indexer)
    seg, _ = time.segments(freq)
    out, val = K.resample_reduce(dev, x, op, seg)
    return _finish(out, val, cell_shape, keep, with_valid)


def select_rolling_resample_op(da, op: str, window: int, time: TimeAxis, window_center: bool = True,
                               window_op: str = "mean", freq: str = "YS", *, device=None, keep=False,
                               with_valid=False, **indexer):
    """gen:128-174: rolling(window).window_op() then select_resample_op (``**indexer`` applies to the ROLLED series)."""
    dev = device or get_device()
    x, cell_shape = _flatten(da, dev)
    rolled = K.rolling_reduce(dev, x, window, window_op, window_center)
    if indexer:
        from .calendar import select_time

        rolled = select_time(rolled, time, device=dev, keep=True, **indexer)
    seg, _ = time.segments(freq)
    out, val = K.resample_reduce(dev, rolled, op, seg)
    return _finish(out, val, cell_shape, keep, with_valid)


def spell_length_statistics(data, threshold, window: int, win_reducer, op: str, spell_reducer, time: TimeAxis, freq: str,
                            min_gap: int = 1, resample_before_rl: bool = True, *, device=None, keep=False,
                            with_valid=False, **indexer):
    """gen:588-686 / 543-585.  window == 1 (the path of maximum_consecutive_dry/wet_days and friends): compare,
    astype(float32), rle_statistics(window=1) fused in ONE kernel pass; window > 1 / min_gap > 1 / several variables go
    through :func:`spell_mask`.  ``spell_reducer`` may be a sequence (-> tuple of results).  ``**indexer`` masks the
    SPELL MASK (gen:558: ``is_in_spell = select_time(is_in_spell, **indexer)``): the run statistics then see NaN steps."""
    if not isinstance(spell_reducer, str):
        return tuple(spell_length_statistics(data, threshold, window, win_reducer, op, sr, time, freq, min_gap,
                                             resample_before_rl, device=device, keep=keep, with_valid=with_valid, **indexer)
                     for sr in spell_reducer)
    sym = get_op(op)
    dev = device or get_device()
    multi = isinstance(data, (list, tuple))
    seg, _ = time.segments(freq)
    if multi:
        flat = [_flatten(d, dev) for d in data]
        x, cell_shape = flat[0]
    else:
        x, cell_shape = _flatten(data, dev)
    if indexer and any(v is not None for k, v in indexer.items() if k != "include_bounds"):
        from .calendar import select_time

        mask = spell_mask([f[0] for f in flat] if multi else x, window, win_reducer, op, threshold, min_gap=min_gap,
                          device=dev, keep=True)
        mask = select_time(mask, time, device=dev, keep=True, **indexer)
        from .run_length import use_ufunc

        one_dim = use_ufunc("from_context", x, freq=None if resample_before_rl else freq)  # NaN steps: rl dispatch
        one_dim = "stat" if one_dim else False  # rl.rle_statistics -> statistics_run_1d
        out, _ = K.run_stats(dev, mask, spell_reducer, 1, seg, cut=resample_before_rl, want_valid=False, one_dim=one_dim)
        val = None
        if with_valid:
            if multi:
                _, val = K.bivariate_count(dev, flat[0][0], flat[1][0], ">", 0.0, ">", 0.0, "all", seg)
            else:
                _, val = K.resample_reduce(dev, x, "count", seg)
        return _finish(out, val, cell_shape, keep, with_valid)
    if window == 1 and min_gap == 1 and not multi:
        cell = _cell_threshold(dev, threshold, data)
        if cell is not None:  # one threshold per grid cell: compared in place (one-row table), fused when the runs are cut
            table, tidx = cell
            if resample_before_rl:
                out, val = K.run_stats_doy(dev, x, sym, table, tidx, spell_reducer, 1, seg)
            else:
                out, _ = K.run_stats(dev, K.compare_doy(dev, x, sym, table, tidx), spell_reducer, 1, seg, cut=False, want_valid=False)
                _, val = K.resample_reduce(dev, x, "count", seg)
            return _finish(out, val, cell_shape, keep, with_valid)
        out, val = K.run_stats(dev, x, spell_reducer, 1, seg, cut=resample_before_rl, fused_op=sym, thresh=float(threshold))
        return _finish(out, val, cell_shape, keep, with_valid)
    if (not multi and min_gap == 1 and resample_before_rl and np.ndim(threshold) == 0 and spell_reducer in
            ("max", "min", "sum", "count", "mean", "std")):
        # window > 1, runs cut at the period edges: spell mask and run statistics in one pass, the mask is never written
        fused = K.spell_run_stats(dev, x, window, win_reducer, sym, float(threshold), spell_reducer, seg)
        if fused is not None:
            return _finish(fused[0], fused[1], cell_shape, keep, with_valid)
    mask = spell_mask([f[0] for f in flat] if multi else x, window, win_reducer, op, threshold, min_gap=min_gap, device=dev,
                      keep=True)
    out, _ = K.run_stats(dev, mask, spell_reducer, 1, seg, cut=resample_before_rl, want_valid=False)
    val = None
    if with_valid:  # valid count of the DATA, not of the mask (every variable must be present)
        if multi:
            _, val = K.bivariate_count(dev, flat[0][0], flat[1][0], ">", 0.0, ">", 0.0, "all", seg)
        else:
            _, val = K.resample_reduce(dev, x, "count", seg)
    return _finish(out, val, cell_shape, keep, with_valid)


def spell_length(data, threshold: float, reducer: str, time: TimeAxis, freq: str, op: str, *, device=None, keep=False,
                 with_valid=False):
    """gen:1204-1252: ``resample_map(compare(data, op, threshold), rl.rle_statistics, reducer, window=1)`` — statistics of
    the spell lengths with the series cut at the period edges; compare and run lengths in one kernel pass."""
    if reducer not in ("max", "min", "mean", "sum"):
        raise ValueError(f"reducer must be one of max, min, mean, sum; got {reducer!r}")
    dev = device or get_device()
    x, cell_shape = _flatten(data, dev)
    seg, _ = time.segments(freq)
    out, val = K.run_stats(dev, x, reducer, 1, seg, cut=True, fused_op=get_op(op), thresh=float(threshold))
    return _finish(out, val, cell_shape, keep, with_valid)


def bivariate_spell_length_statistics(data1, threshold1: float, data2, threshold2: float, window: int, win_reducer, op: str,
                                      spell_reducer, time: TimeAxis, freq: str, min_gap: int = 1,
                                      resample_before_rl: bool = True, *, device=None, keep=False, with_valid=False, **indexer):
    """gen:689-766: spell statistics where BOTH variables fulfil their window condition."""
    return spell_length_statistics([data1, data2], [threshold1, threshold2], window, win_reducer, op, spell_reducer, time,
                                   freq, min_gap, resample_before_rl, device=device, keep=keep, with_valid=with_valid,
                                   **indexer)


def spell_mask(data, window: int, win_reducer: str, op: str, thresh, min_gap: int = 1, weights=None,
               var_reducer: str = "all", *, device=None, keep=False):
    """gen:434-540: boolean mask of the days that are part of a spell.  ``data`` may be a list of variables with one
    (scalar) threshold each; their window conditions are combined with ``var_reducer`` (all / any)."""
    multi = isinstance(data, (list, tuple))
    if multi:
        if not isinstance(thresh, (list, tuple)) or len(thresh) != len(data):
            raise ValueError("When `data` is a sequence, `thresh` must be a sequence of the same length.")
        if var_reducer not in ("all", "any"):
            raise ValueError(f"Unsupported value for var_reducer: {var_reducer}")
    if weights is not None:
        if win_reducer != "mean":
            raise ValueError(f"Argument 'weights' is only supported if 'win_reducer' is 'mean'. Got :  {win_reducer}")
        if len(weights) != window:
            raise ValueError(f"Weights have a different length ({len(weights)}) than the window ({window}).")
    sym = get_op(op)
    dev = device or get_device()
    if multi:
        flat = [_flatten(d, dev) for d in data]
        xs, cell_shape = [f[0] for f in flat], flat[0][1]
        th = [float(t) for t in thresh]
        fast = weights is None and ((win_reducer == "min" and op in (">", ">=", "ge", "gt"))
                                    or (win_reducer == "max" and op in ("`<", "<=", "le", "lt")))
        if window == 1:
            m = K.spell_mask_multi(dev, xs, 1, "min", sym, th, var_reducer)
        elif fast:
            # gen:503-518: the DAILY conditions are combined first, then runs of >= window days are kept (this differs
            # from the general path for var_reducer="any"; the "`<" typo of gen:504 sends "<" to the general path)
            daily = K.spell_mask_multi(dev, xs, 1, "min", sym, th, var_reducer)
            m = K.spell_mask(dev, daily, window, "min", ">=", 1.0)
        else:
            m = K.spell_mask_multi(dev, xs, window, win_reducer, sym, th, var_reducer, weights)
    else:
        x, cell_shape = _flatten(data, dev)
        cell = _cell_threshold(dev, thresh, data)
        if cell is not None:
            # one threshold per grid cell (a DataArray without the time dim in the reference: tests/test_generic.py:754-766).
            # The compare runs against the one-row float64 table; with a window the three steps of gen:519-535 are kept
            # apart: rolling statistic (trailing, NaN until the window is full) -> per-cell compare -> "part of ANY window
            # that satisfies the condition" = trailing rolling max of the condition read w - 1 steps ahead (zeros behind
            # the end of the series).
            table, tidx = cell
            if window == 1:
                m = K.compare_doy(dev, x, sym, table, tidx)
            else:
                T, C_ = x.shape
                # (weights: the dot product of gen:523-524 as a field — xh_rolling_dot — instead of the rolling statistic)
                stat = K.rolling_dot(dev, x, weights) if weights is not None else K.rolling_reduce(dev, x, window, win_reducer, center=False)
                cond = K.compare_doy(dev, stat, sym, table, tidx)
                pad = dev.zeros((T + window - 1, C_), np.float32)
                dev.copy_d2d(pad.ptr, cond.ptr, cond.nbytes)
                anyw = K.rolling_reduce(dev, pad, window, "max", center=False)
                m = dev.wrap(anyw.ptr + (window - 1) * C_ * 4, (T, C_), np.float32)
                m._owner = anyw
        elif window == 1:
            m = K.spell_mask(dev, x, 1, "min", sym, float(thresh))
        else:
            m = K.spell_mask(dev, x, window, win_reducer, sym, float(thresh), weights)
    if min_gap > 1:
        m = K.runs_with_holes(dev, m, 1, None, min_gap)  # rl.runs_with_holes(mask, 1, ~mask, min_gap), gen:537-538
    if keep:
        return m
    return m.get().reshape((m.shape[0],) + tuple(cell_shape)).astype(bool)


def thresholded_events(data, thresh: float, op: str, window: int, thresh_stop=None, op_stop=None, window_stop: int = 1,
                       freq: str | None = None, *, time: TimeAxis | None = None, device=None):
    """gen:1739-1804: event table of run_length.find_events with start condition ``data op thresh`` and stop condition
    ``data op_stop thresh_stop`` (default: the negation of the start condition)."""
    from . import run_length as hrl

    dev = device or get_device()
    x, _ = _flatten(data, dev)
    start = K.compare_map(dev, x, get_op(op), thresh, "maskf")
    if thresh_stop is None and op_stop is None:
        stop = None
    else:
        ts = thresh if thresh_stop is None else thresh_stop
        if op_stop is not None:
            stop = K.compare_map(dev, x, get_op(op_stop), ts, "maskf")
        else:
            stop = K.compare_map(dev, K.compare_map(dev, x, get_op(op), ts, "maskf"), "==", 0.0, "maskf")
    shape = np.shape(data) if not isinstance(data, DeviceArray) else data.shape
    st = start.reshape(*shape)
    sp = None if stop is None else stop.reshape(*shape)
    return hrl.find_events(st, window, sp, window_stop, data=x.reshape(*shape), freq=freq, time=time, device=dev)


def _occurrence(data, threshold, op, time, freq, constrain, device, last):
    from . import run_length as hrl

    sym = get_op(op, constrain)
    dev = device or get_device()
    x, cell_shape = _flatten(data, dev)
    cond = K.spell_mask(dev, x, 1, "min", sym, float(threshold))
    seg, _ = time.segments(freq)
    out, _ = K.run_stats(dev, cond, "last" if last else "first", 1, seg, cut=True, want_valid=False)
    res = hrl._to_coord(out.get(), seg, "dayofyear", time)
    return res.reshape((res.shape[0],) + tuple(cell_shape))


def first_occurrence(data, threshold: float, op: str, time: TimeAxis, freq: str, constrain=None, *, device=None):
    """gen:1108-1152: day of year of the first time step satisfying the condition in each period (NaN if none)."""
    return _occurrence(data, threshold, op, time, freq, constrain, device, last=False)


def last_occurrence(data, threshold: float, op: str, time: TimeAxis, freq: str, constrain=None, *, device=None):
    """gen:1155-1201."""
    return _occurrence(data, threshold, op, time, freq, constrain, device, last=True)


def first_day_threshold_reached(data, *, threshold: float, op: str, after_date: str, time: TimeAxis, window: int = 1,
                                freq: str = "YS", constrain=None, device=None):
    """gen:1555-1608: day of year of the first run of `window` days satisfying the condition on/after `after_date`."""
    from . import run_length as hrl

    sym = get_op(op, constrain)
    dev = device or get_device()
    x, cell_shape = _flatten(data, dev)
    cond = K.spell_mask(dev, x, 1, "min", sym, float(threshold))
    return hrl.first_run_after_date(cond.reshape((x.shape[0],) + tuple(cell_shape)), window, after_date, coord="dayofyear",
                                    time=time, freq=freq, device=dev)


def doymax(da, time: TimeAxis, freq: str = "YS", *, device=None):
    """gen:177-198: day of year of the period maximum."""
    return _doy_extreme(da, time, freq, "argmax", device)


def doymin(da, time: TimeAxis, freq: str = "YS", *, device=None):
    """gen:201-221."""
    return _doy_extreme(da, time, freq, "argmin", device)


def _doy_extreme(da, time, freq, which, device):
    dev = device or get_device()
    x, cell_shape = _flatten(da, dev)
    seg, _ = time.segments(freq)
    idx, _ = K.resample_reduce(dev, x, which, seg)
    std, _ = K.resample_reduce(dev, x, "std", seg, want_valid=False)
    i, s = idx.get(), std.get()
    out = np.full(i.shape, np.nan)
    for p in range(i.shape[0]):
        ok = (i[p] >= 0) & (s[p] != 0)  # tmax.where(std != 0), gen:190-192
        out[p, ok] = time.doy[int(seg[p]) + i[p, ok]]
    return out.reshape((out.shape[0],) + tuple(cell_shape))


def season(data, thresh: float, window: int, op: str, time: TimeAxis, freq: str, mid_date: str | None = None, *,
           device=None):
    """gen:769-853: start / end (as day of year) / length of the season per period.

    cond = compare(data, op, thresh); per period rl.season(cond, window, mid_date) with coord="dayofyear".
    Returns a dict of numpy arrays (P, *cells): "start", "end" (day of year, NaN when undefined), "length" (days).
    """
    from . import run_length as hrl

    sym = get_op(op)
    dev = device or get_device()
    x, cell_shape = _flatten(data, dev)
    cond = K.spell_mask(dev, x, 1, "min", sym, float(thresh))
    return hrl.season(cond.reshape((x.shape[0],) + tuple(cell_shape)), window, mid_date, time=time, freq=freq,
                      coord="dayofyear", device=dev)
