"""Index compositions on the hot path, with the Indicator-level MissingAny mask fused in.

Reference: indices/_simple.py:76-113 (tg_mean & co.), indices/_multivariate.py:1440-1592 (t*10p/t*90p),
indices/_threshold.py:2815-2937 (maximum_consecutive_{dry,wet}_days), Indicator._postprocess
(core/indicator.py:1522-1549) + MissingAny (core/missing.py:318-322).

Every function returns what ``xclim.atmos.<indicator>`` returns after ``check_missing="any"``: a float64 array
(P, *cells) with NaN where the period has a missing/invalid day.  ``mask_missing=False`` gives the raw
``xclim.indices`` result.
"""

from __future__ import annotations

import numpy as np

from . import generic
from . import kernels as K
from ._capi import get_device
from .bootstrapping import percentile_bootstrap
from .calendar import DoyPercentile
from .timeaxis import TimeAxis


import contextvars

# time selection (season= / month= / doy_bounds= / date_bounds=) of the running index call: set by `_indexed`, read by
# `_masked` so that MissingAny expects the SELECTED days only (core/missing.py:118-135)
_INDEXER = contextvars.ContextVar("xclim_amd_indexer", default=None)
_INDEXER_KEYS = ("season", "month", "doy_bounds", "date_bounds", "include_bounds")


def _masked(out, valid, time: TimeAxis, freq, dev, cell_shape, mask_missing):
    if not mask_missing:
        o = out.get()
        return o.reshape((o.shape[0],) + tuple(cell_shape))
    res = K.apply_missing_mask(dev, out, valid, time.expected_count(freq, **(_INDEXER.get() or {}))).get()
    return res.reshape((res.shape[0],) + tuple(cell_shape))


def _indexed(fn, nvars: int):
    """Indicator-level time selection (core/indicator.py: ``da = select_time(da, **indexer)`` on every input before the
    compute, the same indexer handed to the missing-value check): the first `nvars` arguments are masked on the device
    (calendar.select_time) and MissingAny counts against the selected days."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        idx = {k: kwargs.pop(k) for k in _INDEXER_KEYS if k in kwargs}
        if not any(v is not None for k, v in idx.items() if k != "include_bounds"):
            return fn(*args, **kwargs)
        from .calendar import select_time

        time = next((a for a in list(args) + list(kwargs.values()) if isinstance(a, TimeAxis)), None)
        if time is None:
            raise ValueError("a time selection needs the TimeAxis argument")
        dev = kwargs.get("device") or get_device()
        args = list(args)
        for i in range(nvars):
            cells = _cells(args[i])
            args[i] = select_time(args[i], time, device=dev, keep=True, **idx).reshape((len(time),) + tuple(cells))
        token = _INDEXER.set(idx)
        try:
            return fn(*args, **kwargs)
        finally:
            _INDEXER.reset(token)

    return wrapper


def _cells(da):
    return tuple(np.shape(da)[1:]) if not hasattr(da, "dev") else tuple(da.shape[1:])


def _resample_index(da, op, time, freq, device, mask_missing):
    dev = device or get_device()
    out, val = generic.select_resample_op(da, op, time, freq, device=dev, keep=True, with_valid=True)
    return _masked(out, val, time, freq, dev, _cells(da), mask_missing)


def tg_mean(tas, time: TimeAxis, freq: str = "YS", *, device=None, mask_missing=True):
    """indices/_simple.py:76-113."""
    return _resample_index(tas, "mean", time, freq, device, mask_missing)


def tg_max(tas, time, freq="YS", **kw):
    return _resample_index(tas, "max", time, freq, kw.get("device"), kw.get("mask_missing", True))


def tg_min(tas, time, freq="YS", **kw):
    return _resample_index(tas, "min", time, freq, kw.get("device"), kw.get("mask_missing", True))


tx_max = tn_max = tg_max
tx_min = tn_min = tg_min
tx_mean = tn_mean = tg_mean


def _percentile_count(da, per: DoyPercentile, time, freq, op, constrain, device, mask_missing):
    dev = device or get_device()
    cnt, val = generic.threshold_count(da, op, per, time, freq, constrain=constrain, device=dev, keep=True,
                                       with_valid=True)
    return _masked(cnt, val, time, freq, dev, _cells(da), mask_missing)


def _tx90p(tasmax, tasmax_per: DoyPercentile, time: TimeAxis, freq: str = "YS", op: str = ">", *, device=None,
           mask_missing=True):
    """indices/_multivariate.py:1534-1592: days with tasmax > its 90th day-of-year percentile.  ``bootstrap=True``
    (keyword) averages the in-base years over the bootstrap replicas (core/bootstrapping.py)."""
    return _percentile_count(tasmax, tasmax_per, time, freq, op, (">", ">="), device, mask_missing)


tx90p = percentile_bootstrap(_tx90p)
tg90p = tn90p = tx90p


def _tx10p(tasmax, tasmax_per: DoyPercentile, time: TimeAxis, freq: str = "YS", op: str = "<", *, device=None,
           mask_missing=True):
    """indices/_multivariate.py:1596-1650 (``bootstrap=True`` supported like tx90p)."""
    return _percentile_count(tasmax, tasmax_per, time, freq, op, ("<", "<="), device, mask_missing)


tx10p = percentile_bootstrap(_tx10p)
tg10p = tn10p = tx10p


def percentile_exceedance(da, time: TimeAxis, freq: str = "YS", op: str = ">", window: int = 5, per: float = 90.0,
                          alpha: float = 1.0 / 3.0, beta: float = 1.0 / 3.0, *, device=None, mask_missing=True):
    """``tx90p(da, percentile_doy(da, window, per), freq)`` in one pass when the percentile base period is the analysed
    series itself (calendar.py:395-494 + indices/_multivariate.py:1534-1650): identical counts, but the
    (dayofyear, lat, lon) float64 percentile field — 2/3 of the chain's HBM traffic — is never materialised.  Falls back
    to the two-step chain for shapes the fused kernels do not cover (366-day years, other windows, central percentiles
    on multi-year series)."""
    from .calendar import _flatten, percentile_doy

    dev = device or get_device()
    sym = generic.get_op(op)
    x, cell_shape = _flatten(da, dev)
    tb, years, doys = time.doy_table()
    seg, _ = time.segments(freq)
    res = None
    if doys.max() != 366:
        # fused kernels: one contiguous year (sliding window) or a multi-year base period on a calendar without gaps
        # (register top-16 kernel, high / low percentiles); they return None for shapes they do not cover
        period = (np.searchsorted(seg, tb, side="right") - 1).astype(np.int32)  # period of every (year, doy) day
        period[tb < 0] = -1
        res = K.percentile_doy_count(dev, x, tb, window, per, sym, period, len(seg) - 1, alpha, beta)
    if res is None:
        p = percentile_doy(x, time, window, per, alpha, beta, device=dev)
        res = generic.threshold_count(x, sym, p, time, freq, device=dev, keep=True, with_valid=True)
    cnt, val = res
    return _masked(cnt, val, time, freq, dev, cell_shape, mask_missing)


def _percentile_run(da, per: DoyPercentile, stat, window, time, freq, resample_before_rl, op, constrain, dev):
    """Run statistic `stat` (sum = windowed_run_count, count = windowed_run_events, max ...) of ``da op per[dayofyear]``
    per period: (out, valid) device arrays.  resample_before_rl: compare, run lengths and valid counts in ONE pass over x
    and the per-doy table (xh_run_stats_doy); after: compare mask + run statistics across the period edges."""
    from .calendar import _flatten, adjust_doy_calendar, resample_doy_index

    sym = generic.get_op(op, constrain)
    x, cell_shape = _flatten(da, dev)
    doy = adjust_doy_calendar(per, time, dev)
    if doy.data.shape[0] != 1:
        raise ValueError("select one percentile first (DoyPercentile.sel)")
    table = doy.data.reshape(doy.data.shape[1], doy.data.shape[2])
    seg, _ = time.segments(freq)
    tidx = resample_doy_index(doy, time)
    if resample_before_rl:
        out, val = K.run_stats_doy(dev, x, sym, table, tidx, stat, window, seg)
    else:
        mask = K.compare_doy(dev, x, sym, table, tidx)
        out, _ = K.run_stats(dev, mask, stat, window, seg, cut=False, want_valid=False)
        _, val = K.resample_reduce(dev, x, "count", seg)
    return out, val, cell_shape


def percentile_run_stat(da, per: DoyPercentile, op: str, stat: str, window: int, time: TimeAxis, freq: str,
                        resample_before_rl: bool = True, *, constrain=None, device=None):
    """``rl.resample_and_rl(compare(da, op, resample_doy(per, da)), resample_before_rl, rl.<stat function>, window, freq)``
    of the reference (indices/_multivariate.py:137-152, 1778-1793) without the (T, Y, X) float64 threshold and without the
    mask: numpy (P, *cells) float32, no missing-value mask (the building-block form used by xr_adapter.resample_and_rl)."""
    dev = device or get_device()
    out, _, cell_shape = _percentile_run(da, per, stat, window, time, freq, resample_before_rl, op, constrain, dev)
    o = out.get()
    return o.reshape((o.shape[0],) + tuple(cell_shape))


def _percentile_spell(da, per: DoyPercentile, window, time, freq, resample_before_rl, op, constrain, device, mask_missing):
    dev = device or get_device()
    # rl.windowed_run_count: total length of the runs of at least `window` steps (run_length.py:437-488)
    out, val, cell_shape = _percentile_run(da, per, "sum", window, time, freq, resample_before_rl, op, constrain, dev)
    return _masked(out, val, time, freq, dev, cell_shape, mask_missing)


def _warm_spell_duration_index(tasmax, tasmax_per: DoyPercentile, time: TimeAxis, window: int = 6, freq: str = "YS",
                               resample_before_rl: bool = True, op: str = ">", *, device=None, mask_missing=True):
    """indices/_multivariate.py:1693-1793: days that are part of a spell of at least `window` consecutive days with
    tasmax above its day-of-year percentile.  ``bootstrap=True`` as in tx90p (decorated at :1718 in the reference)."""
    return _percentile_spell(tasmax, tasmax_per, window, time, freq, resample_before_rl, op, (">", ">="), device, mask_missing)


def _cold_spell_duration_index(tasmin, tasmin_per: DoyPercentile, time: TimeAxis, window: int = 6, freq: str = "YS",
                               resample_before_rl: bool = True, op: str = "<", *, device=None, mask_missing=True):
    """indices/_multivariate.py:66-152: same with tasmin below its day-of-year percentile (``bootstrap=True``: :68)."""
    return _percentile_spell(tasmin, tasmin_per, window, time, freq, resample_before_rl, op, ("<", "<="), device, mask_missing)


warm_spell_duration_index = percentile_bootstrap(_warm_spell_duration_index)
cold_spell_duration_index = percentile_bootstrap(_cold_spell_duration_index)


def _precip_over(pr, pr_per, time, freq, thresh, op, want, device, mask_missing):
    from .calendar import _flatten, adjust_doy_calendar, resample_doy_index

    dev = device or get_device()
    sym = generic.get_op(op, (">", ">="))
    x, cell_shape = _flatten(pr, dev)
    if isinstance(pr_per, DoyPercentile):  # one value per day of year: resample_doy is fused into the kernel
        doy = adjust_doy_calendar(pr_per, time, dev)
        if doy.data.shape[0] != 1:
            raise ValueError("select one percentile first (DoyPercentile.sel)")
        table = doy.data.reshape(doy.data.shape[1], doy.data.shape[2])
        tidx = resample_doy_index(doy, time)
    else:  # one value per cell (e.g. pr.quantile(q, dim="time"))
        per = pr_per if hasattr(pr_per, "dev") else dev.to_device(np.ascontiguousarray(np.asarray(pr_per, dtype=np.float64).reshape(1, -1)))
        table = per.reshape(1, x.shape[1]) if per.shape != (1, x.shape[1]) else per
        tidx = np.zeros(len(time), dtype=np.int32)
    seg, _ = time.segments(freq)
    cnt, frac, val = K.precip_over_doy(dev, x, sym, float(thresh), table, tidx, seg, want=(want,))
    return _masked(cnt if want == "count" else frac, val, time, freq, dev, cell_shape, mask_missing)


def _days_over_precip_thresh(pr, pr_per, time: TimeAxis, freq: str = "YS", op: str = ">", *, thresh: float, device=None,
                             mask_missing=True):
    """indices/_multivariate.py:1174-1232: wet days (pr op thresh) whose precipitation is also over the percentile
    `pr_per` (a DoyPercentile, or one value per cell).  `thresh` is a float in the units of `pr` (the reference default
    "1 mm/day" is 1/86400 kg m-2 s-1).  ``bootstrap=True`` as in tx90p."""
    return _precip_over(pr, pr_per, time, freq, thresh, op, "count", device, mask_missing)


def _fraction_over_precip_thresh(pr, pr_per, time: TimeAxis, freq: str = "YS", op: str = ">", *, thresh: float, device=None,
                                 mask_missing=True):
    """indices/_multivariate.py:1236-1296: share of the wet-day precipitation that fell on days over the percentile."""
    return _precip_over(pr, pr_per, time, freq, thresh, op, "frac", device, mask_missing)


days_over_precip_thresh = percentile_bootstrap(_days_over_precip_thresh)
fraction_over_precip_thresh = percentile_bootstrap(_fraction_over_precip_thresh)


def _heat_wave(tasmin, tasmax, thresh_tasmin, thresh_tasmax, stat, window, time, freq, op, resample_before_rl, device,
               mask_missing):
    from .calendar import _flatten

    dev = device or get_device()
    sym = generic.get_op(op, (">", ">="))
    tn, cell_shape = _flatten(tasmin, dev)
    tx, cell_shape_x = _flatten(tasmax, dev)
    if tuple(cell_shape) != tuple(cell_shape_x) or tn.shape != tx.shape:
        raise ValueError("tasmin and tasmax must have the same shape")
    seg, _ = time.segments(freq)
    cond = K.spell_mask_multi(dev, [tn, tx], 1, None, sym, [float(thresh_tasmin), float(thresh_tasmax)], "all")
    out, _ = K.run_stats(dev, cond, stat, int(window), seg, cut=bool(resample_before_rl), want_valid=False)
    # MissingAny over both inputs (core/indicator.py `_mask`: logical_or of the per-variable masks)
    _, val = K.bivariate_count(dev, tn, tx, sym, float(thresh_tasmin), sym, float(thresh_tasmax), "all", seg)
    return _masked(out, val, time, freq, dev, cell_shape, mask_missing)


def heat_wave_frequency(tasmin, tasmax, time: TimeAxis, thresh_tasmin: float, thresh_tasmax: float, window: int = 3,
                        freq: str = "YS", op: str = ">", resample_before_rl: bool = True, *, device=None, mask_missing=True):
    """indices/_multivariate.py:640-715: number of runs of at least `window` days with tasmin and tasmax both over their
    thresholds (floats in the units of the data)."""
    return _heat_wave(tasmin, tasmax, thresh_tasmin, thresh_tasmax, "count", window, time, freq, op, resample_before_rl, device,
                      mask_missing)


def heat_wave_max_length(tasmin, tasmax, time: TimeAxis, thresh_tasmin: float, thresh_tasmax: float, window: int = 3,
                         freq: str = "YS", op: str = ">", resample_before_rl: bool = True, *, device=None, mask_missing=True):
    """indices/_multivariate.py:718-794: longest such run (0 when none reaches `window` days)."""
    return _heat_wave(tasmin, tasmax, thresh_tasmin, thresh_tasmax, "max", window, time, freq, op, resample_before_rl, device,
                      mask_missing)


def heat_wave_total_length(tasmin, tasmax, time: TimeAxis, thresh_tasmin: float, thresh_tasmax: float, window: int = 3,
                           freq: str = "YS", op: str = ">", resample_before_rl: bool = True, *, device=None, mask_missing=True):
    """indices/_multivariate.py:797-862: days that belong to such runs."""
    return _heat_wave(tasmin, tasmax, thresh_tasmin, thresh_tasmax, "sum", window, time, freq, op, resample_before_rl, device,
                      mask_missing)


def _spell(da, thresh, op, reducer, time, freq, resample_before_rl, device, mask_missing):
    dev = device or get_device()
    out, val = generic.spell_length_statistics(da, thresh, 1, None, op, reducer, time, freq,
                                               resample_before_rl=resample_before_rl, device=dev, keep=True,
                                               with_valid=True)
    return _masked(out, val, time, freq, dev, _cells(da), mask_missing)


def maximum_consecutive_dry_days(pr, thresh: float, time: TimeAxis, freq: str = "YS", resample_before_rl: bool = True,
                                 op: str = "<", *, device=None, mask_missing=True):
    """indices/_threshold.py:2895-2937; ``thresh`` in the units of ``pr`` (1 mm/day = 1/86400 kg m-2 s-1)."""
    return _spell(pr, thresh, op, "max", time, freq, resample_before_rl, device, mask_missing)


def maximum_consecutive_wet_days(pr, thresh: float, time: TimeAxis, freq: str = "YS", resample_before_rl: bool = True,
                                 op: str = ">", *, device=None, mask_missing=True):
    """indices/_threshold.py:2815-2860."""
    return _spell(pr, thresh, op, "max", time, freq, resample_before_rl, device, mask_missing)


def frost_days(tasmin, time: TimeAxis, thresh: float = 273.15, freq: str = "YS", *, device=None, mask_missing=True):
    """indices/_threshold.py (threshold_count(tasmin, "<", thresh, freq))."""
    dev = device or get_device()
    cnt, val = generic.threshold_count(tasmin, "<", float(thresh), time, freq, device=dev, keep=True, with_valid=True)
    return _masked(cnt, val, time, freq, dev, _cells(tasmin), mask_missing)


# ---- index-level callers of the hot path (indices/_threshold.py, _simple.py, _multivariate.py) -------------------------
# Thin compositions like the reference's own index functions (unit conversion is host / pint work and stays outside:
# thresholds are numbers in the units of the data).  Every one returns (P, *cells) with the MissingAny mask applied.
def _count_index(da, thresh, op, constrain, time, freq, device, mask_missing):
    dev = device or get_device()
    cnt, val = generic.threshold_count(da, op, thresh, time, freq, constrain=constrain, device=dev, keep=True, with_valid=True)
    return _masked(cnt, val, time, freq, dev, _cells(da), mask_missing)


def tx_days_above(tasmax, thresh: float, time: TimeAxis, freq: str = "YS", op: str = ">", *, device=None, mask_missing=True):
    """indices/_threshold.py:2590-2628 (and tn_/tg_days_above :2422-2546)."""
    return _count_index(tasmax, thresh, op, (">", ">="), time, freq, device, mask_missing)


tn_days_above = tg_days_above = tx_days_above


def tx_days_below(tasmax, thresh: float, time: TimeAxis, freq: str = "YS", op: str = "<", *, device=None, mask_missing=True):
    """indices/_threshold.py:2632-2670 (and tn_/tg_days_below :2464-2588)."""
    return _count_index(tasmax, thresh, op, ("<", "<="), time, freq, device, mask_missing)


tn_days_below = tg_days_below = tx_days_below


def ice_days(tasmax, thresh: float, time: TimeAxis, freq: str = "YS", *, device=None, mask_missing=True):
    """indices/_simple.py:412-443: days with tasmax < thresh."""
    return _count_index(tasmax, thresh, "<", None, time, freq, device, mask_missing)


def dry_days(pr, thresh: float, time: TimeAxis, freq: str = "YS", op: str = "<", *, device=None, mask_missing=True):
    """indices/_threshold.py:756-795."""
    return _count_index(pr, thresh, op, ("<", "<="), time, freq, device, mask_missing)


def wetdays(pr, thresh: float, time: TimeAxis, freq: str = "YS", op: str = ">=", *, device=None, mask_missing=True):
    """indices/_threshold.py:2749-2788."""
    return _count_index(pr, thresh, op, (">", ">="), time, freq, device, mask_missing)


def _run_index(da, thresh, op, constrain, stat, window, time, freq, resample_before_rl, device, mask_missing):
    """compare -> resample_and_rl(run statistic): the compare is fused into the run-length kernel."""
    dev = device or get_device()
    sym = generic.get_op(op, constrain)
    from .calendar import _flatten

    x, cell_shape = _flatten(da, dev)
    seg, _ = time.segments(freq)
    cell = generic._cell_threshold(dev, thresh, da)
    if cell is not None:  # one threshold per grid cell (a DataArray threshold in the reference)
        table, tidx = cell
        if resample_before_rl:
            out, val = K.run_stats_doy(dev, x, sym, table, tidx, stat, int(window), seg)
        else:
            out, _ = K.run_stats(dev, K.compare_doy(dev, x, sym, table, tidx), stat, int(window), seg, cut=False, want_valid=False)
            _, val = K.resample_reduce(dev, x, "count", seg)
        return _masked(out, val, time, freq, dev, cell_shape, mask_missing)
    out, val = K.run_stats(dev, x, stat, int(window), seg, cut=bool(resample_before_rl), fused_op=sym, thresh=float(thresh))
    return _masked(out, val, time, freq, dev, cell_shape, mask_missing)


def hot_spell_frequency(tasmax, thresh: float, time: TimeAxis, window: int = 3, freq: str = "YS", op: str = ">",
                        resample_before_rl: bool = True, *, device=None, mask_missing=True):
    """indices/_threshold.py:2291-2351: number of runs of at least `window` days above the threshold."""
    return _run_index(tasmax, thresh, op, (">", ">="), "count", window, time, freq, resample_before_rl, device, mask_missing)


def hot_spell_total_length(tasmax, thresh: float, time: TimeAxis, window: int = 3, freq: str = "YS", op: str = ">",
                           resample_before_rl: bool = True, *, device=None, mask_missing=True):
    """indices/_threshold.py:2232-2288: days that belong to such runs."""
    return _run_index(tasmax, thresh, op, (">", ">="), "sum", window, time, freq, resample_before_rl, device, mask_missing)


def hot_spell_max_length(tasmax, thresh: float, time: TimeAxis, window: int = 1, freq: str = "YS", op: str = ">",
                         resample_before_rl: bool = True, *, device=None, mask_missing=True):
    """indices/_threshold.py:2169-2228: longest run, 0 when shorter than `window` (max_l.where(max_l >= window, 0))."""
    out = _run_index(tasmax, thresh, op, (">", ">="), "max", 1, time, freq, resample_before_rl, device, mask_missing)
    with np.errstate(invalid="ignore"):
        return np.where(out < window, np.where(np.isnan(out), out, 0.0), out)


def hot_spell_max_magnitude(tasmax, thresh: float, time: TimeAxis, window: int = 3, freq: str = "YS",
                            resample_before_rl: bool = True, *, device=None, mask_missing=True):
    """indices/_threshold.py:2019-2066: largest sum of (tasmax - thresh) over a run of at least `window` days above the
    threshold: rl.windowed_max_run_sum of the clipped excess, cut at the period edges unless ``resample_before_rl=False``."""
    from . import run_length as hrl
    from .calendar import _flatten

    dev = device or get_device()
    x, cell_shape = _flatten(tasmax, dev)
    over = K.compare_map(dev, x, ">", float(thresh), "excess")
    seg, _ = time.segments(freq)
    out = hrl.resample_and_rl(over, resample_before_rl, hrl.windowed_max_run_sum, window, freq=freq, time=time, device=dev,
                              keep=True)
    _, val = K.resample_reduce(dev, x, "count", seg)
    return _masked(out, val, time, freq, dev, cell_shape, mask_missing)


def cold_spell_days(tas, thresh: float, time: TimeAxis, window: int = 5, freq: str = "YS-JUL", op: str = "<",
                    resample_before_rl: bool = True, *, device=None, mask_missing=True):
    """indices/_threshold.py:158-213."""
    return _run_index(tas, thresh, op, ("<", "<="), "sum", window, time, freq, resample_before_rl, device, mask_missing)


def cold_spell_frequency(tas, thresh: float, time: TimeAxis, window: int = 5, freq: str = "YS-JUL", op: str = "<",
                         resample_before_rl: bool = True, *, device=None, mask_missing=True):
    """indices/_threshold.py:218-264."""
    return _run_index(tas, thresh, op, ("<", "<="), "count", window, time, freq, resample_before_rl, device, mask_missing)


def maximum_consecutive_tx_days(tasmax, thresh: float, time: TimeAxis, freq: str = "YS", resample_before_rl: bool = True, *,
                                device=None, mask_missing=True):
    """indices/_threshold.py:3003-3060: longest run of days with tasmax > thresh."""
    return _run_index(tasmax, thresh, ">", None, "max", 1, time, freq, resample_before_rl, device, mask_missing)


def maximum_consecutive_frost_days(tasmin, thresh: float, time: TimeAxis, freq: str = "YS-JUL", resample_before_rl: bool = True,
                                   *, device=None, mask_missing=True):
    """indices/_threshold.py:2837-2890: longest run of days with tasmin < thresh."""
    return _run_index(tasmin, thresh, "<", None, "max", 1, time, freq, resample_before_rl, device, mask_missing)


def _degree_days(tas, thresh, op, time, freq, device, mask_missing):
    dev = device or get_device()
    out, val = generic.cumulative_difference(tas, thresh, op, time, freq, device=dev, keep=True, with_valid=True)
    return _masked(out, val, time, freq, dev, _cells(tas), mask_missing)


def growing_degree_days(tas, thresh: float, time: TimeAxis, freq: str = "YS", *, device=None, mask_missing=True):
    """indices/_threshold.py:938-971: sum of (tas - thresh) over the days above the threshold."""
    return _degree_days(tas, thresh, ">", time, freq, device, mask_missing)


def cooling_degree_days(tas, thresh: float, time: TimeAxis, freq: str = "YS", *, device=None, mask_missing=True):
    """indices/_threshold.py:905-935."""
    return _degree_days(tas, thresh, ">", time, freq, device, mask_missing)


def heating_degree_days(tas, thresh: float, time: TimeAxis, freq: str = "YS", *, device=None, mask_missing=True):
    """indices/_threshold.py:2127-2165: sum of (thresh - tas) over the days below the threshold."""
    return _degree_days(tas, thresh, "<", time, freq, device, mask_missing)


def daily_temperature_range(tasmin, tasmax, time: TimeAxis, freq: str = "YS", op: str = "mean", *, device=None,
                            mask_missing=True):
    """indices/_multivariate.py:514-558: `op` of (tasmax - tasmin) per period."""
    dev = device or get_device()
    out, val = generic.diurnal_temperature_range(tasmin, tasmax, op, time, freq, device=dev, keep=True, with_valid=True)
    return _masked(out, val, time, freq, dev, _cells(tasmin), mask_missing)


def daily_temperature_range_variability(tasmin, tasmax, time: TimeAxis, freq: str = "YS", *, device=None, mask_missing=True):
    """indices/_multivariate.py:561-598: mean absolute day-to-day change of the diurnal range."""
    dev = device or get_device()
    out, val = generic.interday_diurnal_temperature_range(tasmin, tasmax, time, freq, device=dev, keep=True, with_valid=True)
    return _masked(out, val, time, freq, dev, _cells(tasmin), mask_missing)


def extreme_temperature_range(tasmin, tasmax, time: TimeAxis, freq: str = "YS", *, device=None, mask_missing=True):
    """indices/_multivariate.py:601-630: max(tasmax) - min(tasmin) per period."""
    dev = device or get_device()
    out, val = generic.extreme_temperature_range(tasmin, tasmax, time, freq, device=dev, keep=True, with_valid=True)
    return _masked(out, val, time, freq, dev, _cells(tasmin), mask_missing)


def growing_season_length(tas, thresh: float, time: TimeAxis, window: int = 6, mid_date: str | None = "07-01",
                          freq: str = "YS", op: str = ">=", *, device=None):
    """indices/_threshold.py:1096-1160 -> generic.season(..., stat="length")."""
    generic.get_op(op, (">", ">="))
    return generic.season(tas, thresh, window, op, time, freq, mid_date, device=device)["length"]


def growing_season_start(tas, thresh: float, time: TimeAxis, mid_date: str | None = "07-01", window: int = 5,
                         freq: str = "YS", op: str = ">=", *, device=None):
    """indices/_threshold.py:975-1026 -> day of year of the season start (NaN without a season)."""
    generic.get_op(op, (">", ">="))
    return generic.season(tas, thresh, window, op, time, freq, mid_date, device=device)["start"]


def growing_season_end(tas, thresh: float, time: TimeAxis, mid_date: str | None = "07-01", window: int = 5,
                       freq: str = "YS", op: str = ">=", *, device=None):
    """indices/_threshold.py:1029-1092 -> day of year of the season end."""
    generic.get_op(op, (">", ">="))
    return generic.season(tas, thresh, window, op, time, freq, mid_date, device=device)["end"]


def _spell_index(pr, thresh, window, win_op, op, spell_reducer, time, freq, resample_before_rl, device, mask_missing):
    dev = device or get_device()
    indexer = _INDEXER.get() or {}
    out, val = generic.spell_length_statistics(pr, float(thresh), int(window), win_op, op, spell_reducer, time, freq,
                                               resample_before_rl=resample_before_rl, device=dev, keep=True, with_valid=True,
                                               **indexer)
    return _masked(out, val, time, freq, dev, _cells(pr), mask_missing)


def dry_spell_frequency(pr, time: TimeAxis, thresh: float = 1.0, window: int = 3, freq: str = "YS",
                        resample_before_rl: bool = True, op: str = "sum", *, device=None, mask_missing=True):
    """indices/_threshold.py:3314-3382: number of periods of at least `window` days whose accumulated (op="sum") or
    maximal (op="max") daily amount stays under `thresh`.  `pr` is the DAILY AMOUNT in the units of `thresh` (the
    reference converts the flux to mm/day first: host work, the caller's)."""
    return _spell_index(pr, thresh, window, op, "<", "count", time, freq, resample_before_rl, device, mask_missing)


def dry_spell_total_length(pr, time: TimeAxis, thresh: float = 1.0, window: int = 3, op: str = "sum", freq: str = "YS",
                           resample_before_rl: bool = True, *, device=None, mask_missing=True):
    """indices/_threshold.py:3385-3454: days that belong to such dry spells."""
    return _spell_index(pr, thresh, window, op, "<", "sum", time, freq, resample_before_rl, device, mask_missing)


def dry_spell_max_length(pr, time: TimeAxis, thresh: float = 1.0, window: int = 1, op: str = "sum", freq: str = "YS",
                         resample_before_rl: bool = True, *, device=None, mask_missing=True):
    """indices/_threshold.py:3457-3522: the longest such dry spell."""
    return _spell_index(pr, thresh, window, op, "<", "max", time, freq, resample_before_rl, device, mask_missing)


def wet_spell_frequency(pr, time: TimeAxis, thresh: float = 1.0, window: int = 3, freq: str = "YS",
                        resample_before_rl: bool = True, op: str = "sum", *, device=None, mask_missing=True):
    """indices/_threshold.py:3525-3593: the same with ``>=``."""
    return _spell_index(pr, thresh, window, op, ">=", "count", time, freq, resample_before_rl, device, mask_missing)


def wet_spell_total_length(pr, time: TimeAxis, thresh: float = 1.0, window: int = 3, op: str = "sum", freq: str = "YS",
                           resample_before_rl: bool = True, *, device=None, mask_missing=True):
    """indices/_threshold.py:3596-3664."""
    return _spell_index(pr, thresh, window, op, ">=", "sum", time, freq, resample_before_rl, device, mask_missing)


def wet_spell_max_length(pr, time: TimeAxis, thresh: float = 1.0, window: int = 1, op: str = "sum", freq: str = "YS",
                         resample_before_rl: bool = True, *, device=None, mask_missing=True):
    """indices/_threshold.py:3667-3735."""
    return _spell_index(pr, thresh, window, op, ">=", "max", time, freq, resample_before_rl, device, mask_missing)


# ---- Indicator-level time selections on every index of this module -------------------------------------------------------
_TWO_INPUTS = {"heat_wave_frequency", "heat_wave_max_length", "heat_wave_total_length", "daily_temperature_range",
               "daily_temperature_range_variability", "extreme_temperature_range"}
# the spell indices hand the selection to spell_length_statistics, which masks the SPELL MASK (gen:558), not the input
_MASK_LEVEL = {"dry_spell_frequency", "dry_spell_total_length", "dry_spell_max_length", "wet_spell_frequency",
               "wet_spell_total_length", "wet_spell_max_length"}
for _n, _f in list(globals().items()):
    if callable(_f) and not _n.startswith("_") and getattr(_f, "__module__", None) == __name__ and not isinstance(_f, type):
        globals()[_n] = _indexed(_f, 0 if _n in _MASK_LEVEL else (2 if _n in _TWO_INPUTS else 1))
del _n, _f
