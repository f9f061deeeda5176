"""Oracle: percentile bootstrapping (reference: src/xclim/core/bootstrapping.py:81-282).  TEST INFRASTRUCTURE ONLY.

Restated for the percentile-exceedance count indices (tx90p family): `da` (T, ...) daily, percentiles computed on the
base period [base_start_year, base_end_year] with percentile_doy.  For every year group of `da` that lies inside the
base period, the base series is rebuilt n-1 times with that year's block replaced by each OTHER base year
(`build_bootstrap_year_da`, :235-282, materialised here with numpy), percentile_doy is recomputed on each replica and
the index of that year is averaged over the replicas (:203); years outside the base period use the original
percentile.
"""

from __future__ import annotations

import numpy as np

from . import calendar as ocal
from . import generic as ogen
from .timeutil import OTime, groups


def bootstrap_freq(freq: str) -> str:
    """bootstrapping.py:214-223 `_get_bootstrap_freq` for start-anchored offsets."""
    f = freq.upper()
    if (f.startswith("YS") or f.startswith("AS") or f.startswith("QS")) and "-" in f:
        return "YS-" + f.split("-")[1]
    return "YS"


def _replace_block(base, bloc_idx, src_idx, time: OTime):
    """bootstrapping.py:253-279: write `source` into the block, with the reference's length rules."""
    out = base.copy()
    nb, ns = len(bloc_idx), len(src_idx)
    if ns < 360 and ns < nb:
        return out  # partial first/last group of an anchored frequency: replica left unchanged
    if ns == nb:
        out[bloc_idx] = base[src_idx]
    elif nb == 365:  # source is a leap year: convert_calendar("noleap") drops Feb 29
        keep = ~((time.month[src_idx] == 2) & (time.day[src_idx] == 29))
        out[bloc_idx] = base[src_idx[keep]]
    elif nb == 366:  # source is a non-leap year: 366_day calendar with NaN on Feb 29
        vals = np.full((366,) + base.shape[1:], np.nan, dtype=base.dtype)
        feb29 = int(np.nonzero((time.month[bloc_idx] == 2) & (time.day[bloc_idx] == 29))[0][0])
        vals[np.arange(366) != feb29] = base[src_idx]
        out[bloc_idx] = vals
    elif nb < 365:
        out[bloc_idx] = base[src_idx[:nb]]
    else:
        raise NotImplementedError
    return out


def bootstrap_exceedance(da, time: OTime, base_years, freq, op=">", window=5, per=90.0, alpha=1.0 / 3.0, beta=1.0 / 3.0,
                         index_fn=None, per_out=None):
    """tx90p-style index with bootstrap=True.  Returns float64 (P, ...) counts (non-integer inside the base period).
    `index_fn(x, per, per_doys, time)` replaces the exceedance count (e.g. days_over_precip_thresh).
    `per_out = (per_da, per_doys)`: the percentile the caller supplied, used for the years outside the overlap of `da`
    with the base period (bootstrapping.py:196-199); default: percentile_doy of the overlap (the two coincide when the
    percentile was computed from `da` itself)."""
    da = np.asarray(da)
    y0, y1 = base_years
    in_base = (time.year >= y0) & (time.year <= y1)
    bidx = np.nonzero(in_base)[0]
    if len(bidx) == len(time):
        raise KeyError("`bootstrap` is unnecessary when all years are overlapping between reference and studied periods")
    if len(bidx) == 0:
        raise KeyError("`bootstrap` is unnecessary when no year overlap between reference and studied periods.")
    base = da[bidx]
    tbase = time.isel(bidx)
    per_da, per_doys = per_out if per_out is not None else ocal.percentile_doy(base, tbase, window, per, alpha, beta)
    bfreq = bootstrap_freq(freq)
    base_groups = groups(tbase, bfreq)
    base_year_labels = set(tbase.year.tolist())
    constrain = (">", ">=") if op in (">", ">=") else ("<", "<=")

    def index(x, t, p, doys):
        if index_fn is not None:
            return np.asarray(index_fn(x, p[..., 0], doys, t)).astype(np.float64)
        thresh = ocal.resample_doy(p[..., 0], doys, t)
        return ogen.threshold_count(x, op, thresh, t, freq, constrain=constrain).astype(np.float64)

    acc = []
    for label, gidx in groups(time, bfreq):
        x_g, t_g = da[gidx], time.isel(gidx)
        year_label = label.year if hasattr(label, "year") else label[0]
        if year_label in base_year_labels:
            # block of this group inside the base series
            bloc = np.nonzero(np.isin(bidx, gidx))[0]
            vals = []
            for lab_s, sidx in base_groups:
                ys = lab_s.year if hasattr(lab_s, "year") else lab_s[0]
                if ys == year_label:
                    continue
                rep = _replace_block(base, bloc, sidx, tbase)
                p_i, doys_i = ocal.percentile_doy(rep, tbase, window, per, alpha, beta)
                vals.append(index(x_g, t_g, p_i, doys_i))
            acc.append(np.mean(np.stack(vals, axis=0), axis=0))
        else:
            acc.append(index(x_g, t_g, per_da, per_doys))
    return np.concatenate(acc, axis=0)
