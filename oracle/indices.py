"""Oracle: the index compositions on the benchmark path + MissingAny.  TEST INFRASTRUCTURE ONLY.

Reference: indices/_simple.py:76-113 (tg_mean), indices/_multivariate.py:1534-1592 (tx90p),
indices/_threshold.py:2895-2937 (maximum_consecutive_dry_days), core/missing.py:201-220, 318-322 (MissingAny).
"""

from __future__ import annotations

import numpy as np

from . import calendar as ocal
from . import generic as ogen
from .timeutil import OTime, days_in_period, groups


def missing_any(da, time: OTime, freq):
    """core/missing.py:318-322: period is missing iff count(notnull) != expected number of days (missing.py:64-160)."""
    da = np.asarray(da)
    valid = np.stack([(~np.isnan(da[idx])).sum(axis=0) for _, idx in groups(time, freq)], axis=0)
    expected = days_in_period(time, freq)
    return valid != expected.reshape((-1,) + (1,) * (da.ndim - 1))


def apply_missing(out, da, time: OTime, freq):
    """Indicator._postprocess (core/indicator.py:1522-1549): out.where(~mask) -> float64 NaN where missing."""
    mask = missing_any(da, time, freq)
    res = np.asarray(out).astype(np.float64)
    res[mask] = np.nan
    return res


def tg_mean(tas, time: OTime, freq="YS"):
    """indices/_simple.py:113: select_resample_op(tas, op="mean", freq)."""
    return ogen.select_resample_op(tas, "mean", time, freq)


def tx90p(tasmax, tasmax_per, per_doys, time: OTime, freq="YS", op=">"):
    """indices/_multivariate.py:1584-1592: thresh = resample_doy(per, tasmax); threshold_count(tasmax, op, thresh).

    `tasmax_per` is the (ndoy, ...) float64 output of percentile_doy for one percentile; the compare is evaluated
    in float64 because the threshold array is float64 (numpy promotion).
    """
    thresh = ocal.resample_doy(tasmax_per, per_doys, time)
    return ogen.threshold_count(tasmax, op, thresh, time, freq, constrain=(">", ">="))


def tx10p(tasmax, tasmax_per, per_doys, time: OTime, freq="YS", op="<"):
    """indices/_multivariate.py:1596-1650 (same as tx90p with the "<" family of operators)."""
    thresh = ocal.resample_doy(tasmax_per, per_doys, time)
    return ogen.threshold_count(tasmax, op, thresh, time, freq, constrain=("<", "<="))


def maximum_consecutive_dry_days(pr, thresh, time: OTime, freq="YS", resample_before_rl=True):
    """indices/_threshold.py:2925-2937: spell_length_statistics(pr, thresh, 1, None, "<", "max", freq).

    `thresh` must be a python float already in the units of `pr` (convert_units_to is host work)."""
    return ogen.spell_length_statistics(pr, float(thresh), 1, None, "<", "max", time, freq, resample_before_rl)


def warm_spell_duration_index(tasmax, tasmax_per, per_doys, time: OTime, window=6, freq="YS", resample_before_rl=True, op=">"):
    """indices/_multivariate.py:1779-1793: above = compare(tasmax, op, resample_doy(per)); windowed_run_count."""
    from . import run_length as rl

    thresh = ocal.resample_doy(tasmax_per, per_doys, time)
    above = ogen.compare(tasmax, op, thresh, constrain=(">", ">="))
    return rl.resample_and_rl(above, resample_before_rl, rl.windowed_run_count, time=time, freq=freq, window=window)


def cold_spell_duration_index(tasmin, tasmin_per, per_doys, time: OTime, window=6, freq="YS", resample_before_rl=True, op="<"):
    """indices/_multivariate.py:139-152."""
    from . import run_length as rl

    thresh = ocal.resample_doy(tasmin_per, per_doys, time)
    below = ogen.compare(tasmin, op, thresh, constrain=("<", "<="))
    return rl.resample_and_rl(below, resample_before_rl, rl.windowed_run_count, time=time, freq=freq, window=window)


# ---- index-level compositions (indices/_threshold.py etc.), units stripped --------------------------------------------
def count_days(da, op, thresh, time: OTime, freq, constrain=None):
    """tx_days_above / dry_days / wetdays / ice_days ...: threshold_count (indices/_threshold.py:2626-2628)."""
    return ogen.threshold_count(da, op, thresh, time, freq, constrain=constrain)


def run_index(da, op, thresh, fn, window, time: OTime, freq, resample_before_rl=True, constrain=None):
    """hot/cold spell indices: compare -> rl.resample_and_rl(cond, before, fn, window, freq) (indices/_threshold.py:2341-2349)."""
    from . import run_length as rl

    cond = ogen.compare(da, op, thresh, constrain)
    f = {"events": rl.windowed_run_events, "count": rl.windowed_run_count}[fn]
    return rl.resample_and_rl(cond, resample_before_rl, f, time=time, freq=freq, window=window)


def longest_run_index(da, op, thresh, window, time: OTime, freq, resample_before_rl=True):
    """hot_spell_max_length (indices/_threshold.py:2213-2227): longest_run, zeroed below `window`."""
    from . import run_length as rl

    cond = ogen.compare(da, op, thresh)
    max_l = rl.resample_and_rl(cond, resample_before_rl, rl.longest_run, time=time, freq=freq)
    return np.where(max_l >= window, max_l, 0)


# ---- wet-day percentile indices and the heat-wave family --------------------------------------------
def _wet_percentile_threshold(pr_per, per_doys, thresh, time: OTime):
    """indices/_multivariate.py:1223-1228: tp = pr_per.where(pr_per > thresh, thresh) (NaN -> thresh), broadcast to the
    time axis when it carries a dayofyear coordinate (per_doys is not None)."""
    pr_per = np.asarray(pr_per, dtype=np.float64)
    with np.errstate(invalid="ignore"):
        tp = np.where(pr_per > thresh, pr_per, float(thresh))
    if per_doys is not None:
        tp = ocal.resample_doy(tp, per_doys, time)
    return tp


def days_over_precip_thresh(pr, pr_per, per_doys, time: OTime, thresh, freq="YS", op=">"):
    """indices/_multivariate.py:1220-1232.  `pr_per`: (ndoy, ...) with `per_doys`, or (...) per cell (per_doys None)."""
    tp = _wet_percentile_threshold(pr_per, per_doys, thresh, time)
    return ogen.threshold_count(pr, op, tp, time, freq, constrain=(">", ">="))


def fraction_over_precip_thresh(pr, pr_per, per_doys, time: OTime, thresh, freq="YS", op=">"):
    """indices/_multivariate.py:1281-1296: sum of pr over the percentile threshold / sum of pr on wet days."""
    pr = np.asarray(pr)
    tp = _wet_percentile_threshold(pr_per, per_doys, thresh, time)
    constrain = (">", ">=")
    wet = np.where(ogen.compare(pr, op, float(thresh), constrain), pr, pr.dtype.type(0))
    big = np.where(ogen.compare(pr, op, tp, constrain), pr, pr.dtype.type(0))
    total = ogen._resample_reduce(wet, time, freq, lambda g: g.sum(axis=0))
    over = ogen._resample_reduce(big, time, freq, lambda g: g.sum(axis=0))
    with np.errstate(invalid="ignore", divide="ignore"):
        return over / total


def _heat_wave_cond(tasmin, tasmax, thresh_tasmin, thresh_tasmax, op):
    constrain = (">", ">=")
    return ogen.compare(tasmin, op, float(thresh_tasmin), constrain) & ogen.compare(tasmax, op, float(thresh_tasmax), constrain)


def heat_wave_frequency(tasmin, tasmax, time: OTime, thresh_tasmin, thresh_tasmax, window=3, freq="YS", op=">",
                        resample_before_rl=True):
    """indices/_multivariate.py:701-715: windowed_run_events of (tasmin op a) & (tasmax op b)."""
    from . import run_length as rl

    cond = _heat_wave_cond(tasmin, tasmax, thresh_tasmin, thresh_tasmax, op)
    return rl.resample_and_rl(cond, resample_before_rl, rl.windowed_run_events, time=time, freq=freq, window=window)


def heat_wave_max_length(tasmin, tasmax, time: OTime, thresh_tasmin, thresh_tasmax, window=3, freq="YS", op=">",
                         resample_before_rl=True):
    """indices/_multivariate.py:781-794: rle_statistics(reducer="max", window)."""
    from . import run_length as rl

    cond = _heat_wave_cond(tasmin, tasmax, thresh_tasmin, thresh_tasmax, op)
    return rl.resample_and_rl(cond, resample_before_rl, rl.rle_statistics, time=time, freq=freq, reducer="max", window=window)


def heat_wave_total_length(tasmin, tasmax, time: OTime, thresh_tasmin, thresh_tasmax, window=3, freq="YS", op=">",
                           resample_before_rl=True):
    """indices/_multivariate.py:849-862: windowed_run_count."""
    from . import run_length as rl

    cond = _heat_wave_cond(tasmin, tasmax, thresh_tasmin, thresh_tasmax, op)
    return rl.resample_and_rl(cond, resample_before_rl, rl.windowed_run_count, time=time, freq=freq, window=window)


def hot_spell_max_magnitude(tasmax, thresh, time: OTime, window=3, freq="YS", resample_before_rl=True):
    """indices/_threshold.py:2056-2065: over = (tasmax - thresh).clip(0); resample_and_rl(over, ..., windowed_max_run_sum)."""
    from . import run_length as rl

    tasmax = np.asarray(tasmax)
    with np.errstate(invalid="ignore"):
        over = np.clip(tasmax - tasmax.dtype.type(thresh), 0, None)
    return rl.resample_and_rl(over, resample_before_rl, rl.windowed_max_run_sum, window, time=time, freq=freq)
