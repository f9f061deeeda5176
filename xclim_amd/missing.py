"""Host mirror of the missing-value checks (reference: core/missing.py:64-160 expected_count, :201-298 MissingBase.is_valid /
__call__, :311-336 MissingAny / MissingSomeButNotAll, :340-393 MissingTwoSteps, :396-448 MissingWMO, :451-481 MissingPct,
:484-512 AtLeastNValid) for a DAILY source.

The index functions of :mod:`xclim_amd.indices` fuse the default check (``MissingAny``) into their kernels (a valid-count
side output + ``xh_apply_missing_mask``); this module is the stand-alone form of ``xclim.core.missing``'s shortcut functions
(``missing_any(da, freq)`` ...).  Every method reduces to the per-period count of valid steps (``xh_resample_reduce``
"count") — the WMO criterion adds the longest run of invalid steps inside each month (``xh_run_stats`` on the NaN mask, runs
cut at the month edges like the reference's ``resample_map(~valid, "time", "MS", rl.longest_run)``) — followed by
arithmetic on the small ``(periods, cells)`` tables.  The two-step methods first decide at ``subfreq`` (months) and call a
period missing when any of its months is, or when the series does not hold all the months of the period
(``MissingAny()(miss.where(~miss), freq, src_timestep="MS")``, core/missing.py:387-392).
"""

from __future__ import annotations

import numpy as np

from . import kernels as K
from ._capi import get_device
from .calendar import _flatten
from .timeaxis import TimeAxis, parse_freq

__all__ = ["expected_count", "missing_any", "missing_some_but_not_all", "missing_wmo", "missing_pct", "at_least_n_valid"]


def _has(indexer) -> bool:
    return bool(indexer) and any(v is not None for k, v in indexer.items() if k != "include_bounds")


def expected_count(time: TimeAxis, freq: str | None, **indexer) -> np.ndarray:
    """core/missing.py:64-160 for a daily source: days of every full period (the selected days with an indexer);
    ``freq=None``: the days from the first to the last step of the series (:124-127, :157-159)."""
    if freq is not None:
        return time.expected_count(freq, **indexer)
    o = time.ordinal()
    n = int(o[-1] - o[0]) + 1
    if not _has(indexer):
        return np.array([n], dtype=np.int32)
    from .calendar import select_time_mask

    synth = TimeAxis.daily(f"{int(time.year[0]):04d}-{int(time.month[0]):02d}-{int(time.day[0]):02d}", n, time.calendar)
    return np.array([int(select_time_mask(synth, **indexer).sum())], dtype=np.int32)


def _selected(da, time, dev, indexer):
    x, cell_shape = _flatten(da, dev)
    if _has(indexer):
        from .calendar import select_time

        x = select_time(x, time, device=dev, keep=True, **indexer)
    return x, cell_shape


def _segments(time: TimeAxis, freq):
    return np.array([0, len(time)], dtype=np.int64) if freq is None else time.segments(freq)[0]


def _valid_and_expected(da, freq, time, dev, indexer):
    """(valid (P, *cells) int, expected (P, 1...) int, the selected device field, cell shape)"""
    x, cell_shape = _selected(da, time, dev, indexer)
    _, valid = K.resample_reduce(dev, x, "count", _segments(time, freq))
    exp = expected_count(time, freq, **indexer).reshape((-1,) + (1,) * len(cell_shape))
    return valid.get().reshape((valid.shape[0],) + tuple(cell_shape)).astype(np.int64), exp.astype(np.int64), x, cell_shape


def missing_any(da, freq: str | None, time: TimeAxis, *, device=None, **indexer) -> np.ndarray:
    """core/missing.py:318-322: True where a period holds fewer valid (non-NaN) steps than expected.  With ``**indexer``
    the series is masked by select_time first and only the selected days are expected (core/missing.py:118-135)."""
    v, exp, _, _ = _valid_and_expected(da, freq, time, device or get_device(), indexer)
    return v != exp


def missing_some_but_not_all(da, freq: str | None, time: TimeAxis, *, device=None, **indexer) -> np.ndarray:
    """core/missing.py:325-336: some, but not all, of a period's steps are missing."""
    v, exp, _, _ = _valid_and_expected(da, freq, time, device or get_device(), indexer)
    return ~((v == exp) | (v == 0))


def _months_to_periods(miss_m: np.ndarray, time: TimeAxis, freq: str | None, indexer) -> np.ndarray:
    """The second step of MissingTwoSteps.__call__ (core/missing.py:387-392): ``MissingAny()(miss.where(~miss), freq,
    src_timestep="MS", **indexer)`` on the monthly mask — a month counts as valid when it is present in the series,
    selected by the indexer and not missing; a period is missing when its valid months are not ALL the (selected) months of
    the full period (expected_count with a monthly source: a synthetic monthly series over the full periods, :118-135)."""
    from .calendar import select_time_mask

    _, mstarts = time.segments("MS")
    months = TimeAxis([y for y, _ in mstarts], [m for _, m in mstarts], np.ones(len(mstarts), dtype=np.int64), time.calendar)
    if _has(indexer):
        if any(indexer.get(k) is not None for k in ("doy_bounds", "date_bounds")):
            raise NotImplementedError("two-step missing methods with doy_bounds / date_bounds: day selections of a MONTHLY mask "
                                      "(use month= / season=, or subfreq=None)")
        sel = select_time_mask(months, **indexer)
    else:
        sel = np.ones(len(months), dtype=bool)
    ok = (~miss_m) & sel.reshape((-1,) + (1,) * (miss_m.ndim - 1))
    if freq is None:
        seg = np.array([0, len(months)], dtype=np.int64)
        exp = np.array([int(sel.sum())])   # start_time .. end_time = the first .. last month of the series itself (:124-127)
    else:
        seg, pstarts = months.segments(freq)
        base, _ = parse_freq(freq)
        nmon = {"M": 1, "Q": 3, "Y": 12}.get(base)
        if nmon is None:
            raise NotImplementedError(f"two-step missing methods resample months to {freq!r}: a month-based frequency is needed")
        exp = []
        for (y, m) in pstarts:
            full = TimeAxis([y + (m - 1 + i) // 12 for i in range(nmon)], [(m - 1 + i) % 12 + 1 for i in range(nmon)],
                            np.ones(nmon, dtype=np.int64), time.calendar)
            exp.append(int(select_time_mask(full, **indexer).sum()) if _has(indexer) else nmon)
        exp = np.asarray(exp)
    nvalid = np.stack([ok[a:b].sum(axis=0) for a, b in zip(seg[:-1], seg[1:])], axis=0)
    return nvalid != exp.reshape((-1,) + (1,) * (miss_m.ndim - 1))


def _two_steps(first, da, freq, time, subfreq, dev, indexer):
    """MissingTwoSteps.__call__ (core/missing.py:352-393): decide at ``subfreq or freq``; when that is finer than ``freq``,
    merge the sub-periods with the "any" rule."""
    sub = subfreq or freq
    same = sub == freq or (sub is not None and freq is not None and parse_freq(sub) == parse_freq(freq))
    if not same:
        if parse_freq(sub) != ("M", 1):
            raise NotImplementedError(f"subfreq={subfreq!r}: the HIP path decides the first step at months ('MS') or at freq itself")
        if freq is not None and parse_freq(freq)[0] not in ("M", "Q", "Y"):
            raise ValueError(f"The target resampling frequency cannot be finer than the first-step frequency. Got : {sub} > {freq}.")
    miss = first(sub)
    return miss if same else _months_to_periods(miss, time, freq, indexer)


def missing_pct(da, freq: str | None, time: TimeAxis, tolerance: float = 0.1, subfreq: str | None = None, *, device=None,
                **indexer) -> np.ndarray:
    """core/missing.py:451-481: the share of missing steps of a period reaches ``tolerance``."""
    if not 0 <= tolerance <= 1:
        raise ValueError(f"Options {{'tolerance': {tolerance}, 'subfreq': {subfreq!r}}} are not valid for MissingPct.")
    dev = device or get_device()

    def first(f):
        v, exp, _, _ = _valid_and_expected(da, f, time, dev, indexer)
        with np.errstate(invalid="ignore", divide="ignore"):
            return ((exp - v) / exp) >= tolerance     # (a period without expected steps: 0 / 0 -> NaN -> False)

    return _two_steps(first, da, freq, time, subfreq, dev, indexer)


def at_least_n_valid(da, freq: str | None, time: TimeAxis, n: int = 20, subfreq: str | None = None, *, device=None,
                     **indexer) -> np.ndarray:
    """core/missing.py:484-512: fewer than ``n`` valid steps in a period (the expected count plays no role)."""
    if not n > 0:
        raise ValueError(f"Options {{'n': {n}, 'subfreq': {subfreq!r}}} are not valid for AtLeastNValid.")
    dev = device or get_device()

    def first(f):
        v, _, _, _ = _valid_and_expected(da, f, time, dev, indexer)
        return v < n

    return _two_steps(first, da, freq, time, subfreq, dev, indexer)


def missing_wmo(da, freq: str | None, time: TimeAxis, nm: int = 11, nc: int = 5, *, device=None, **indexer) -> np.ndarray:
    """core/missing.py:396-448: a month is missing with ``nm`` or more missing days or a run of ``nc`` or more consecutive
    missing days; a longer period is missing when any of its months is."""
    if not (nm < 31 and nc < 31):
        raise ValueError(f"Options {{'nm': {nm}, 'nc': {nc}}} are not valid for MissingWMO.")
    dev = device or get_device()

    def first(f):   # f == "MS"
        v, exp, x, cell_shape = _valid_and_expected(da, f, time, dev, indexer)
        # ~valid: 1 where the (selected) field is NaN — x != x is the one compare that is True for NaN only
        invalid = K.compare_map(dev, x, "!=", x, "maskf")
        run, _ = K.run_stats(dev, invalid, "max", 1, time.segments(f)[0], cut=True, want_valid=False)
        longest = run.get().reshape(v.shape)
        return ((exp - v) >= nm) | (longest >= nc)

    return _two_steps(first, da, freq, time, "MS", dev, indexer)
