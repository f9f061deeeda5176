"""Oracle: generic index kernels (reference: src/xclim/indices/generic.py).  TEST INFRASTRUCTURE ONLY.

Time on axis 0.  Thresholds are unit-less numbers (unit handling is host/pint work, out of scope).
"""

from __future__ import annotations

import operator
import warnings

import numpy as np

from . import run_length as rl
from .timeutil import OTime, groups

# gen:40-75 binary_ops
BINARY_OPS = {">": "gt", "<": "lt", ">=": "ge", "<=": "le", "==": "eq", "!=": "ne"}


def get_op(op: str, constrain=None):
    """gen:255-298."""
    if op in BINARY_OPS:
        binary_op = BINARY_OPS[op]
    elif op in BINARY_OPS.values():
        binary_op = op
    else:
        raise ValueError(f"Operation `{op}` not recognized.")
    if constrain:
        allowed = list(constrain) + [BINARY_OPS[c] for c in constrain if c in BINARY_OPS]
        if op not in allowed:
            raise ValueError(f"Operation `{op}` not permitted for indice.")
    return getattr(operator, f"__{binary_op}__")


def compare(left, op, right, constrain=None):
    """gen:301-326: python operator on arrays; NaN compares False (True for !=); numpy promotion rules apply."""
    with np.errstate(invalid="ignore"):
        return get_op(op, constrain)(left, right)


def _resample_reduce(arr, time: OTime, freq, func):
    return np.stack([func(arr[idx]) for _, idx in groups(time, freq)], axis=0)


def threshold_count(da, op, threshold, time: OTime, freq, constrain=None):
    """gen:329-361: (compare * 1).resample(time=freq).sum("time") -> int64."""
    if constrain is None:
        constrain = (">", "<", ">=", "<=")
    c = compare(da, op, threshold, constrain) * 1
    return _resample_reduce(c, time, freq, lambda g: g.sum(axis=0))


def domain_count(da, low, high, time: OTime, freq):
    """gen:364-392: ((da > low) & (da <= high)) * 1 -> resample.sum."""
    with np.errstate(invalid="ignore"):
        c = ((da > low) & (da <= high)) * 1
    return _resample_reduce(c, time, freq, lambda g: g.sum(axis=0))


def count_occurrences(da, threshold, op, time: OTime, freq, constrain=None):
    """gen:960-999: compare(...).resample.sum."""
    c = compare(da, op, threshold, constrain) * 1
    return _resample_reduce(c, time, freq, lambda g: g.sum(axis=0))


def count_level_crossings(low, high, threshold, time: OTime, freq, op_low="<", op_high=">="):
    """gen:913-957."""
    c = (compare(low, op_low, threshold, ("<", "<=")) & compare(high, op_high, threshold, (">", ">="))) * 1
    return _resample_reduce(c, time, freq, lambda g: g.sum(axis=0))


def bivariate_count_occurrences(v1, v2, t1, t2, time: OTime, freq, op1, op2, var_reducer):
    """gen:1002-1073."""
    c1, c2 = compare(v1, op1, t1), compare(v2, op2, t2)
    c = ((c1 & c2) if var_reducer == "all" else (c1 | c2)) * 1
    return _resample_reduce(c, time, freq, lambda g: g.sum(axis=0))


def get_daily_events(da, threshold, op, constrain=None):
    """gen:395-431: 1 where the condition holds, 0 where not, NaN where da is NaN."""
    da = np.asarray(da)
    events = compare(da, op, threshold, constrain) * 1
    return np.where(np.isnan(da), np.nan, events)


def diurnal_temperature_range(low_data, high_data, reducer, time: OTime, freq="YS"):
    """gen:1076-1105: reducer((high - low).resample(time=freq)); NaN-skipping xarray reducers."""
    dtr = np.asarray(high_data) - np.asarray(low_data)
    return _resample_reduce(dtr, time, freq, lambda g: _nanreduce(g, reducer))


def interday_diurnal_temperature_range(low_data, high_data, time: OTime, freq="YS"):
    """gen:1360-1385: abs((high - low).diff("time")).resample(time=freq).mean(); diff drops the first day, so the
    first period holds one value less."""
    dtr = np.asarray(high_data) - np.asarray(low_data)
    vdtr = np.abs(np.diff(dtr, axis=0))
    out = []
    for _, idx in groups(time, freq):
        idx = idx[idx >= 1] - 1  # position in the diff'ed series (time coordinate = the later day)
        out.append(_nanreduce(vdtr[idx], "mean"))
    return np.stack(out)


def extreme_temperature_range(low_data, high_data, time: OTime, freq="YS"):
    """gen:1388-1414: high.resample.max() - low.resample.min()."""
    hi = _resample_reduce(np.asarray(high_data), time, freq, lambda g: _nanreduce(g, "max"))
    lo = _resample_reduce(np.asarray(low_data), time, freq, lambda g: _nanreduce(g, "min"))
    return hi - lo


def thresholded_statistics(data, op, threshold, reducer, time: OTime, freq):
    """gen:1278-1320: getattr(data.where(cond).resample(time=freq), reducer)()."""
    data = np.asarray(data)
    masked = np.where(compare(data, op, threshold), data, np.nan).astype(data.dtype)
    return _resample_reduce(masked, time, freq, lambda g: _nanreduce(g, reducer))


def temperature_sum(data, op, threshold, time: OTime, freq):
    """gen:1323-1357."""
    data = np.asarray(data)
    cond = compare(data, op, threshold, ("<", "<=", ">", ">="))
    direction = -1 if op in ["<", "<=", "lt", "le"] else 1
    d = np.where(cond, data - threshold, np.nan).astype(data.dtype)
    return direction * _resample_reduce(d, time, freq, lambda g: _nanreduce(g, "sum"))


def cumulative_difference(data, threshold, op, time: OTime, freq):
    """gen:1514-1552."""
    data = np.asarray(data)
    if op in ["<", "<=", "lt", "le"]:
        diff = np.clip(threshold - data, 0, None)
    elif op in [">", ">=", "gt", "ge"]:
        diff = np.clip(data - threshold, 0, None)
    else:
        raise NotImplementedError(f"Condition not supported: '{op}'.")
    return _resample_reduce(diff.astype(data.dtype), time, freq, lambda g: _nanreduce(g, "sum"))


def _nanreduce(g, op, acc_dtype=np.float64):
    """xarray's default float reductions (skipna=True) on one group; accumulation dtype stated: fp64."""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        if op == "count":
            return (~np.isnan(g)).sum(axis=0)
        if g.shape[0] == 0:
            shape = g.shape[1:]
            return np.zeros(shape, g.dtype) if op in ("sum", "integral") else np.full(shape, np.nan, g.dtype)
        if op in ("sum", "integral"):
            return np.nansum(g, axis=0, dtype=acc_dtype).astype(g.dtype)
        if op == "mean":
            return np.nanmean(g, axis=0, dtype=acc_dtype).astype(g.dtype)
        if op == "min":
            return np.nanmin(g, axis=0)
        if op == "max":
            return np.nanmax(g, axis=0)
        if op == "std":
            return np.nanstd(g, axis=0, dtype=acc_dtype).astype(g.dtype)
        if op == "var":
            return np.nanvar(g, axis=0, dtype=acc_dtype).astype(g.dtype)
        if op in ("argmin", "argmax"):
            allnan = np.isnan(g).all(axis=0)
            filled = np.where(allnan, 0, g)
            r = (np.nanargmin if op == "argmin" else np.nanargmax)(filled, axis=0)
            return np.where(allnan, -1, r)
    raise ValueError(op)


def select_resample_op(da, op, time: OTime, freq="YS"):
    """gen:83-125 without indexer: da.resample(time=freq).<op>(dim="time")."""
    return _resample_reduce(np.asarray(da), time, freq, lambda g: _nanreduce(g, op))


def rolling(da, window, op, center=True):
    """da.rolling(time=window, center=center).<op>() with xarray defaults (min_periods = window): gen:166-170.

    Window of step t: [t - w//2, t + w - 1 - w//2] when centred, [t - w + 1, t] otherwise.  Any NaN or an
    incomplete window gives NaN.  fp64 accumulation.
    """
    da = np.asarray(da)
    T = da.shape[0]
    left = window // 2 if center else window - 1
    out = np.full(da.shape, np.nan, dtype=da.dtype)
    for t in range(T):
        a, b = t - left, t - left + window - 1
        if a < 0 or b >= T:
            continue
        g = da[a : b + 1].astype(np.float64)
        if op in ("sum", "integral"):
            r = g.sum(axis=0)
        elif op == "mean":
            r = g.sum(axis=0) / window
        elif op == "min":
            r = g.min(axis=0)
        elif op == "max":
            r = g.max(axis=0)
        elif op == "std":
            r = g.std(axis=0)
        elif op == "var":
            r = g.var(axis=0)
        else:
            raise ValueError(op)
        out[t] = r  # NaN propagates through sum/min/max
    return out


def select_rolling_resample_op(da, op, window, time: OTime, window_center=True, window_op="mean", freq="YS"):
    """gen:128-174."""
    return select_resample_op(rolling(da, window, window_op, window_center), op, time, freq)


def spell_mask(data, window, win_reducer, op, thresh, min_gap=1, weights=None, var_reducer="all"):
    """gen:434-540.  `data` may be a list of variables (with a list of thresholds): the reference concatenates them
    along a "variable" dimension and reduces the per-variable condition with all / any at the point restated here."""
    if isinstance(data, (list, tuple)):
        if not isinstance(thresh, (list, tuple)) or len(thresh) != len(data):
            raise ValueError("When `data` is a sequence, `thresh` must be a sequence of the same length.")
        m = _spell_mask_nogap([np.asarray(d) for d in data], window, win_reducer, op, list(thresh), weights, var_reducer)
    else:
        m = _spell_mask_nogap([np.asarray(data)], window, win_reducer, op, [thresh], weights, "all")
    if min_gap > 1:
        m = rl.runs_with_holes(m, 1, ~m, min_gap).astype(bool)  # gen:537-538
    return m


def _spell_mask_nogap(datas, window, win_reducer, op, threshs, weights=None, var_reducer="all"):
    def reduce_vars(masks):
        st = np.stack(masks)
        return st.all(axis=0) if var_reducer == "all" else st.any(axis=0)

    if weights is not None:
        if win_reducer != "mean":
            raise ValueError(f"Argument 'weights' is only supported if 'win_reducer' is 'mean'. Got :  {win_reducer}")
        if len(weights) != window:
            raise ValueError(f"Weights have a different length ({len(weights)}) than the window ({window}).")
    T = datas[0].shape[0]
    if window == 1:
        return reduce_vars([compare(d, op, th) for d, th in zip(datas, threshs)])
    if weights is None and ((win_reducer == "min" and op in [">", ">=", "ge", "gt"])
                            or (win_reducer == "max" and op in ["`<", "<=", "le", "lt"])):
        # gen:503-518 (the literal "`<" typo of gen:504 is kept: "<" takes the general path)
        mask = reduce_vars([compare(d, op, th) for d, th in zip(datas, threshs)])
        cs_s = rl.cumsum_reset(mask)
        with np.errstate(invalid="ignore"):
            cs_s = rl.where_nan(cs_s, rl.shift0(mask.astype(np.float64), -1, 0) == 0)
            v = rl.where_nan(cs_s, cs_s >= window)
        v = np.where(mask > 0, v, 0)
        # bfill along time
        idx = np.where(~np.isnan(v), np.arange(T).reshape((-1,) + (1,) * (v.ndim - 1)), T - 1)
        idx = np.minimum.accumulate(idx[::-1], axis=0)[::-1]
        filled = np.take_along_axis(v, idx, axis=0)
        with np.errstate(invalid="ignore"):
            return filled > 0
    # general path gen:519-535
    masks = []
    for data, thresh in zip(datas, threshs):
        pad = np.concatenate([data, np.full((window,) + data.shape[1:], np.nan, dtype=data.dtype)], axis=0)
        if weights is not None:
            sv = np.full(pad.shape, np.nan)
            w = np.asarray(weights, dtype=np.float64)
            for t in range(window - 1, pad.shape[0]):
                sv[t] = np.tensordot(w, pad[t - window + 1 : t + 1].astype(np.float64), axes=(0, 0))
            spell_value = sv.astype(np.float32)
        else:
            spell_value = rolling(pad, window, win_reducer, center=False)
        masks.append(compare(spell_value, op, thresh))
    mask = reduce_vars(masks)
    msum = rolling(mask.astype(np.float64), window, "sum", center=False)
    with np.errstate(invalid="ignore"):
        is_in = msum >= 1
    is_in = rl.shift0(is_in, -(window - 1), False)
    return is_in[:T]


def spell_length_statistics(data, thresh, window, win_reducer, op, spell_reducer, time: OTime, freq,
                            resample_before_rl=True, min_gap=1, **indexer):
    """gen:543-585 / 588-686 (and the bivariate form gen:689-766 when data / thresh are lists) without indexer:
    mask -> float32 -> resample_and_rl(rle_statistics, window=1)."""
    mask = spell_mask(data, window, win_reducer, op, thresh, min_gap=min_gap).astype(np.float32)
    if indexer:  # gen:558: the time selection masks the SPELL MASK (NaN outside)
        from . import calendar as ocal

        mask = ocal.select_time(mask, time, **indexer)
    return rl.resample_and_rl(mask, resample_before_rl, rl.rle_statistics, time=time, freq=freq, reducer=spell_reducer,
                              window=1)


def spell_length(data, threshold, reducer, time: OTime, freq, op):
    """gen:1204-1252: cond = compare(data, op, threshold); rle_statistics(reducer, window=1) mapped over the periods."""
    from . import run_length as rl

    cond = compare(data, op, threshold)
    return rl.resample_and_rl(cond, True, rl.rle_statistics, time=time, freq=freq, reducer=reducer, window=1)
