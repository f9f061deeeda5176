"""Oracle: the missing-value methods of xclim.core.missing for a daily source.  TEST INFRASTRUCTURE ONLY.

Restates /root/reference/src/xclim/core/missing.py: expected_count :64-160, MissingBase.is_valid / __call__ :201-298,
MissingAny :311-322, MissingSomeButNotAll :325-336, MissingTwoSteps.__call__ :352-393, MissingWMO :396-448, MissingPct
:451-481, AtLeastNValid :484-512 — with plain numpy loops over the resampling groups (oracle/timeutil.py), independent of
xclim_amd.  Pinned by the reference's own known answers (tests/test_missing.py:166-285, ported in
tests/test_oracle_reference_answers.py).
"""

from __future__ import annotations

import numpy as np

from . import calendar as ocal
from .timeutil import OTime, days_in_period, groups

_ML = [31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31]


def _has(indexer):
    return any(v is not None for k, v in indexer.items() if k != "include_bounds")


def _full_series(time: OTime, y0, m0, ndays):
    """A complete daily series of `ndays` from the first of month (y0, m0), in the calendar of `time` (missing.py:121-127)."""
    if time.index is not None:
        return OTime.standard(f"{y0:04d}-{m0:02d}-01", ndays)
    ml = [30] * 12 if time.calendar == "360_day" else _ML
    skip = sum(ml[:m0 - 1])
    return OTime.noleap(y0, skip + ndays, time.calendar).isel(slice(skip, None))


def expected_count(time: OTime, freq, **indexer):
    """missing.py:64-160, daily source.  freq=None: from the first to the last step."""
    if freq is None:
        if time.index is not None:
            n = (time.index[-1] - time.index[0]).days + 1
            full = OTime.standard(time.index[0], n)
        else:
            ml = [30] * 12 if time.calendar == "360_day" else _ML
            ord_ = lambda y, m, d: y * sum(ml) + sum(ml[:m - 1]) + d  # noqa: E731
            n = ord_(time.year[-1], time.month[-1], time.day[-1]) - ord_(time.year[0], time.month[0], time.day[0]) + 1
            skip = sum(ml[:time.month[0] - 1]) + time.day[0] - 1
            full = OTime.noleap(int(time.year[0]), skip + n, time.calendar).isel(slice(skip, None))
        return np.array([n if not _has(indexer) else int(ocal.select_time_mask(full, **indexer).sum())])
    base = days_in_period(time, freq)
    if not _has(indexer):
        return base
    gs = groups(time, freq)
    lab = gs[0][0]
    y0, m0 = (lab.year, lab.month) if hasattr(lab, "year") else lab
    if time.index is not None and freq.upper().startswith(("YE", "QE", "ME", "A-", "Q-", "Y-")):
        raise NotImplementedError("end-anchored frequencies with an indexer")
    mask = ocal.select_time_mask(_full_series(time, y0, m0, int(base.sum())), **indexer)
    edges = np.concatenate([[0], np.cumsum(base)])
    return np.array([int(mask[a:b].sum()) for a, b in zip(edges[:-1], edges[1:])])


def _valid(da, time, indexer):
    """MissingBase.is_valid (missing.py:201-220): select_time(da, **indexer).notnull()"""
    da = np.asarray(da)
    sel = ocal.select_time(da, time, **indexer) if _has(indexer) else da
    return ~np.isnan(sel)


def _idx_groups(time, freq):
    return [np.arange(len(time))] if freq is None else [idx for _, idx in groups(time, freq)]


def _sums(valid, time, freq):
    return np.stack([valid[idx].sum(axis=0) for idx in _idx_groups(time, freq)], axis=0)


def _count(time, freq, indexer, ndim):
    return expected_count(time, freq, **indexer).reshape((-1,) + (1,) * (ndim - 1))


def missing_any(da, time: OTime, freq, **indexer):
    valid = _valid(da, time, indexer)
    return _sums(valid, time, freq) != _count(time, freq, indexer, valid.ndim)


def missing_some_but_not_all(da, time: OTime, freq, **indexer):
    valid = _valid(da, time, indexer)
    s = _sums(valid, time, freq)
    return ~((s == _count(time, freq, indexer, valid.ndim)) | (s == 0))


def _month_axis(time: OTime):
    """(year, month) of every month from the first to the last one of the series — the time axis of a mask resampled at "MS"
    (empty months inside the span included)."""
    return [lab if isinstance(lab, tuple) else (lab.year, lab.month) for lab, _ in groups(time, "MS")]


def _nmon(freq):
    f = freq.upper().replace("AS", "YS")
    if f in ("MS", "ME", "M"):
        return 1, 0
    mon = ["JAN", "FEB", "MAR", "APR", "MAY", "JUN", "JUL", "AUG", "SEP", "OCT", "NOV", "DEC"]
    if f.startswith("YS"):
        return 12, (mon.index(f.split("-")[1]) if "-" in f else 0)
    if f.startswith("QS"):
        return 3, (mon.index(f.split("-")[1]) if "-" in f else 0) % 3
    raise NotImplementedError(freq)


def _month_selected(months, indexer):
    """select_time on a monthly series: month= / season= look at the month number of each step"""
    mon = np.array([m for _, m in months])
    if not _has(indexer):
        return np.ones(len(mon), dtype=bool)
    if indexer.get("month") is not None:
        want = indexer["month"]
        return np.isin(mon, [want] if np.isscalar(want) else list(want))
    if indexer.get("season") is not None:
        names = {"DJF": (12, 1, 2), "MAM": (3, 4, 5), "JJA": (6, 7, 8), "SON": (9, 10, 11)}
        s = indexer["season"]
        return np.isin(mon, [m for k in ([s] if isinstance(s, str) else s) for m in names[k]])
    raise NotImplementedError("day selections of a monthly mask")


def _any_of_months(miss_m, time: OTime, freq, indexer):
    """MissingTwoSteps.__call__ second step (missing.py:387-392): MissingAny()(miss.where(~miss), freq, src_timestep="MS", **indexer)"""
    months = _month_axis(time)
    valid = (~miss_m) & _month_selected(months, indexer).reshape((-1,) + (1,) * (miss_m.ndim - 1))
    if freq is None:
        return valid.sum(axis=0, keepdims=True) != _month_selected(months, indexer).sum()
    nmon, off = _nmon(freq)
    key = np.array([(y * 12 + m - 1 - off) // nmon for y, m in months])
    out = []
    for k in range(key.min(), key.max() + 1):
        ms = k * nmon + off
        full = [((ms + i) // 12, (ms + i) % 12 + 1) for i in range(nmon)]
        out.append(valid[key == k].sum(axis=0) != _month_selected(full, indexer).sum())
    return np.stack(out, axis=0)


def _two_steps(first, da, time, freq, subfreq, indexer):
    sub = subfreq or freq
    miss = first(sub)
    if sub != freq:
        assert sub.upper() == "MS"
        miss = _any_of_months(miss, time, freq, indexer)
    return miss


def missing_pct(da, time: OTime, freq, tolerance=0.1, subfreq=None, **indexer):
    valid = _valid(da, time, indexer)

    def first(f):
        count = _count(time, f, indexer, valid.ndim).astype(np.float64)
        with np.errstate(invalid="ignore", divide="ignore"):
            return ((count - _sums(valid, time, f)) / count) >= tolerance

    return _two_steps(first, da, time, freq, subfreq, indexer)


def at_least_n_valid(da, time: OTime, freq, n=20, subfreq=None, **indexer):
    valid = _valid(da, time, indexer)
    return _two_steps(lambda f: _sums(valid, time, f) < n, da, time, freq, subfreq, indexer)


def _longest_true_run(col):
    best = cur = 0
    for v in col:
        cur = cur + 1 if v else 0
        best = max(best, cur)
    return best


def missing_wmo(da, time: OTime, freq, nm=11, nc=5, **indexer):
    valid = _valid(da, time, indexer)

    def first(f):
        cond1 = (_count(time, f, indexer, valid.ndim) - _sums(valid, time, f)) >= nm
        runs = []
        for idx in _idx_groups(time, f):
            blk = (~valid[idx]).reshape(len(idx), -1)
            runs.append(np.array([_longest_true_run(blk[:, c]) for c in range(blk.shape[1])]).reshape(valid.shape[1:]))
        return cond1 | (np.stack(runs, axis=0) >= nc)

    return _two_steps(first, da, time, freq, "MS", indexer)
