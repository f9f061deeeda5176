"""GPU parity at the API level: the host mirrors of the reference modules (xclim_amd.{calendar,generic,run_length,
indices,utils,sdba}) against the oracle, plus the reference's own known answers run through the HIP path."""

import os
import warnings

import numpy as np
import pytest

from oracle import calendar as ocal
from oracle import generic as ogen
from oracle import indices as oidx
from oracle import quantile as oq
from oracle import run_length as orl
from oracle import sdba as osdba
from oracle.timeutil import OTime
from xclim_amd import generic as xgen
from xclim_amd import indices as xi
from xclim_amd import run_length as xrl
from xclim_amd import sdba as xsdba
from xclim_amd import utils as xutils
from xclim_amd.calendar import percentile_doy
from xclim_amd.timeaxis import TimeAxis

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.npz"))


def _temp(rng, T, shape, nan_frac=0.0):
    t = np.arange(T).reshape((T,) + (1,) * len(shape))
    x = (288 + 12 * np.sin(2 * np.pi * (t - 100) / 365) + rng.normal(0, 3, (T,) + shape)).astype(np.float32)
    if nan_frac:
        x[rng.random(x.shape) < nan_frac] = np.nan
    return x


def _axes(start, T, calendar="standard"):
    if calendar == "standard":
        return TimeAxis.daily(start, T), OTime.standard(start, T)
    return TimeAxis.daily(start, T, calendar), OTime.noleap(int(start[:4]), T, calendar)


def _accept_plane_nearest_ties(bad, got, x, gcoord, hist_q, af, kind, limit=6):
    """Windowed groups SHARE samples, so the extreme nodes of the groups d - 1 and d + 1 are often the same value: two nodes of
    the (hist_q, group) plane at exactly the same distance of a query on row d.  scipy's cKDTree returns either (230 : 170 in 400
    constructed ties), the kernel the lower row.  Clears the entries of `bad` (T, *cells) whose value IS that of a node at the
    minimal distance; anything else stays flagged.  hist_q, af: (G, nq, *cells); gcoord: the integer group coordinate (1 .. G)."""
    assert bad.sum() <= limit, f"{int(bad.sum())} mismatches"
    G = hist_q.shape[0]
    for idx in np.argwhere(bad):
        t, cell = int(idx[0]), tuple(int(i) for i in idx[1:])
        hq = hist_q[(slice(None), slice(None)) + cell].astype(np.float64)
        a = af[(slice(None), slice(None)) + cell].astype(np.float64)
        xv = float(x[(t,) + cell])
        k = np.arange(-3, 4)
        rows = (int(gcoord[t]) - 1 + k) % G
        d2 = (hq[rows] - xv) ** 2 + (k ** 2)[:, None]
        cand = (xv + a[rows] if kind == "+" else xv * a[rows])[d2 <= d2.min() * (1 + 1e-12)]
        if len(cand) >= 2 and np.isclose(cand, got[(t,) + cell], rtol=1e-6).any():
            bad[(t,) + cell] = False
    return bad


@pytest.mark.parametrize("calendar,T", [("standard", 1461), ("standard", 366), ("noleap", 1095), ("noleap", 365)])
@pytest.mark.parametrize("per", [90.0, [10.0, 50.0]])
def test_percentile_doy_full(dev, rng, calendar, T, per):
    """percentile_doy including the drop-366 / re-interpolate step (cal:484-485) and its attrs."""
    x = _temp(rng, T, (6, 5), nan_frac=0.002)
    ta, ot = _axes("2000-01-01", T, calendar)
    p = percentile_doy(x, ta, window=5, per=per, device=dev)
    exp, doys = ocal.percentile_doy(x, ot, 5, per)
    np.testing.assert_array_equal(p.dayofyear, doys)
    np.testing.assert_allclose(p.values(), exp, rtol=1e-12, atol=0, equal_nan=True)
    assert p.attrs["window"] == 5 and p.attrs["alpha"] == 1 / 3 and "percentile_doy" in p.attrs["history"]
    assert p.attrs["climatology_bounds"][0] == "2000-01-01"


@pytest.mark.parametrize("calendar,T,freq", [("standard", 1461, "YS"), ("standard", 800, "MS"), ("noleap", 1095, "QS-DEC"),
                                             ("noleap", 365, "YS")])
def test_tx90p_tx10p_indicator_level(dev, rng, calendar, T, freq):
    x = _temp(rng, T, (7, 9), nan_frac=0.001)
    ta, ot = _axes("2000-01-01", T, calendar)
    for per, fn, ofn in ((90.0, xi.tx90p, oidx.tx90p), (10.0, xi.tx10p, oidx.tx10p)):
        p = percentile_doy(x, ta, window=5, per=per, device=dev)
        got = fn(x, p, ta, freq=freq, device=dev)
        pe, doys = ocal.percentile_doy(x, ot, 5, per)
        exp = oidx.apply_missing(ofn(x, pe[..., 0], doys, ot, freq), x, ot, freq)
        assert got.dtype == np.float64 and got.shape == exp.shape
        np.testing.assert_array_equal(got, exp)
        raw = fn(x, p, ta, freq=freq, device=dev, mask_missing=False)
        np.testing.assert_array_equal(raw, ofn(x, pe[..., 0], doys, ot, freq))


def test_tg_mean_and_friends(dev, rng):
    x = _temp(rng, 800, (4, 4), nan_frac=0.001)
    ta, ot = _axes("1999-12-01", 800)
    for freq in ("YS", "MS"):
        got = xi.tg_mean(x, ta, freq, device=dev)
        exp = oidx.apply_missing(oidx.tg_mean(x, ot, freq), x, ot, freq)
        np.testing.assert_allclose(got, exp, rtol=1e-6, equal_nan=True)
        np.testing.assert_array_equal(np.isnan(got), np.isnan(exp))
    np.testing.assert_array_equal(xi.tg_max(x, ta, "YS", device=dev, mask_missing=False),
                                  ogen.select_resample_op(x, "max", ot, "YS"))


@pytest.mark.parametrize("before", [True, False])
def test_reference_cdd_known_answers(dev, before):
    """reference tests/test_indices.py:2354-2381 through the HIP path: 10 ; 26 (resample before) vs 30 (after)."""
    ta = TimeAxis.daily("2000-07-01", 365)
    thr = 1.0 / 86400.0
    a = np.zeros(365, np.float32) + 10
    a[5:15] = 0
    assert xi.maximum_consecutive_dry_days(a[:, None], thr, ta, "ME", before, device=dev, mask_missing=False)[0, 0] == 10
    a = np.zeros(365, np.float32) + 10
    a[5:35] = 0
    out = xi.maximum_consecutive_dry_days(a[:, None], thr, ta, "ME", before, device=dev, mask_missing=False)
    assert out[0, 0] == (26 if before else 30)


def test_reference_txp_known_answers(dev):
    """reference tests/test_indices.py:2529-2625: t*10p Jan 0 / Jun 5 ; t*90p (per=10) Jan 30 / Feb 29 / Jun 25."""
    tas = np.arange(366).astype(np.float32)
    ta = TimeAxis.daily("2000-01-01", 366)
    t10 = percentile_doy(tas[:, None], ta, per=10.0, device=dev)
    assert len(t10.dayofyear) == 366
    tas[175:180] = 1
    out = xi.tx10p(tas[:, None], t10, ta, "MS", device=dev, mask_missing=False)[:, 0]
    assert out[0] == 0 and out[5] == 5
    out = xi.tx90p(tas[:, None], t10, ta, "MS", device=dev, mask_missing=False)[:, 0]
    assert out[0] == 30 and out[1] == 29 and out[5] == 25


def test_reference_percentile_doy_known_answers(dev):
    """reference tests/test_calendar.py:83-104."""
    tas = np.arange(365).astype(np.float32)
    ta = TimeAxis.daily("2001-01-01", 365)
    p = percentile_doy(np.stack([tas, tas], 1), ta, window=5, per=50, device=dev)
    assert p.values()[2, 0, 0] == 2
    tas[1] = np.nan
    p = percentile_doy(np.stack([tas, tas], 1), ta, window=5, per=50, device=dev)
    assert p.values()[2, 0, 0] == 2.5


def test_reference_run_length_known_answers(dev):
    """reference tests/test_run_length.py:166-296, 356-424 through the HIP path."""
    v = np.ones(365, np.float32)
    v[35] = 0
    ta = TimeAxis.daily("2000-01-01", 365)
    m = v[:, None]
    assert xrl.rle_statistics(m, "min", 1, freq="YS", time=ta, device=dev)[0, 0] == 35
    assert xrl.rle_statistics(m, "mean", 36, freq="YS", time=ta, device=dev)[0, 0] == 329
    assert xrl.rle_statistics(m, "std", 1, freq="YS", time=ta, device=dev)[0, 0] == 147
    ta7 = TimeAxis.daily("2000-07-01", 365)
    after = xrl.rle_statistics(m, "max", 1, freq="ME", time=ta7, device=dev)[:, 0]
    assert after[0] == 35 and after[1] == 365 - 35 - 1
    before = xrl.resample_and_rl(m, True, xrl.rle_statistics, "max", 1, freq="ME", time=ta7, device=dev)[:, 0]
    assert before[0] == 31 and before[1] == 26
    a = np.zeros((50, 1), np.float32)
    a[4:7] = 1
    a[34:45] = 1
    for index in ("first", "last"):
        assert xrl.windowed_run_events(a, 3, index=index, device=dev)[0] == 2
        assert xrl.windowed_run_count(a, 3, index=index, device=dev)[0] == 14
    t = np.zeros((60, 2), np.float32)
    t[30:40] = 2
    np.testing.assert_array_equal(xrl.first_run(t, 1, device=dev), [30, 30])
    np.testing.assert_array_equal(xrl.last_run(t, 1, device=dev), [39, 39])
    t[0] = 2
    tj = TimeAxis.daily("2000-01-01", 60)
    np.testing.assert_array_equal(xrl.first_run(t, 1, freq="MS", time=tj, device=dev), [[0, 0], [0, 0]])
    np.testing.assert_array_equal(xrl.last_run(t, 1, freq="MS", time=tj, device=dev), [[30, 30], [8, 8]])
    with pytest.raises(ValueError):
        xrl.rle_statistics(m, "max", 1, freq="YS", ufunc_1dim=True, time=ta, device=dev)


def test_run_length_mirror_vs_oracle(dev, rng):
    T, shape = 500, (5, 6)
    m = (rng.random((T,) + shape) < 0.6)
    ta, ot = _axes("2001-03-01", T)
    np.testing.assert_array_equal(xrl.rle(m, device=dev), orl.rle(m.astype(np.float32)))
    np.testing.assert_array_equal(xrl._cumsum_reset(m, index="first", device=dev), orl.cumsum_reset(m.astype(np.float32), "first"))
    np.testing.assert_array_equal(xrl.longest_run(m, device=dev), orl.longest_run(m))
    np.testing.assert_array_equal(xrl.windowed_run_count(m, 1, device=dev), orl.windowed_run_count(m, 1))
    np.testing.assert_array_equal(xrl.windowed_run_count(m, 4, freq="MS", time=ta, device=dev),
                                  orl.windowed_run_count(m, 4, time=ot, freq="MS"))
    np.testing.assert_array_equal(xrl.first_run(m, 3, freq="MS", time=ta, device=dev), orl.first_run(m, 3, time=ot, freq="MS"))


def test_generic_mirror_vs_oracle(dev, rng):
    T = 730
    x = _temp(rng, T, (8,), nan_frac=0.01)
    ta, ot = _axes("2000-01-01", T)
    np.testing.assert_array_equal(xgen.threshold_count(x, ">", 290.0, ta, "MS", device=dev), ogen.threshold_count(x, ">", 290.0, ot, "MS"))
    np.testing.assert_array_equal(xgen.threshold_count(x, ">", np.float64(290.1), ta, "MS", device=dev),
                                  ogen.threshold_count(x, ">", np.float64(290.1), ot, "MS"))
    full = rng.normal(289, 2, x.shape)
    np.testing.assert_array_equal(xgen.threshold_count(x, "<", full, ta, "YS", device=dev), ogen.threshold_count(x, "<", full, ot, "YS"))
    np.testing.assert_array_equal(xgen.domain_count(x, 285.0, 295.0, ta, "YS", device=dev),
                                  ogen.domain_count(x, np.float32(285), np.float32(295), ot, "YS"))
    np.testing.assert_array_equal(xgen.count_occurrences(x, 290.0, "!=", ta, "YS", device=dev), ogen.count_occurrences(x, 290.0, "!=", ot, "YS"))
    got = xgen.select_rolling_resample_op(x, "max", 5, ta, True, "mean", "YS", device=dev)
    np.testing.assert_allclose(got, ogen.select_rolling_resample_op(x, "max", 5, ot, True, "mean", "YS"), rtol=1e-6)
    with pytest.raises(ValueError, match="not permitted"):
        xgen.threshold_count(x, "==", 1.0, ta, "YS", device=dev)
    with pytest.raises(ValueError, match="not recognized"):
        xgen.threshold_count(x, "=>", 1.0, ta, "YS", device=dev)


@pytest.mark.parametrize("calendar,start,nyears,base,freq", [("noleap", "2000-01-01", 7, (2001, 2004), "YS"),
                                                             ("noleap", "2000-01-01", 6, (2000, 2003), "MS"),
                                                             ("standard", "1998-01-01", 7, (1999, 2002), "YS"),
                                                             ("standard", "1999-01-01", 5, (1999, 2001), "QS-DEC")])
def test_percentile_bootstrap(dev, rng, calendar, start, nyears, base, freq):
    """core/bootstrapping.py: in-base years are averaged over n-1 replicas built through virtual time maps; must equal
    the oracle, which materialises every replica like `build_bootstrap_year_da` (incl. the 365 <-> 366 rules)."""
    from oracle import bootstrapping as oboot

    T = 365 * nyears + (sum(1 for y in range(int(start[:4]), int(start[:4]) + nyears) if y % 4 == 0) if calendar == "standard" else 0)
    x = _temp(rng, T, (3, 4), nan_frac=0.003)
    ta, ot = _axes(start, T, calendar)
    from xclim_amd.calendar import percentile_doy as pdoy

    b0 = int(np.nonzero(ta.year >= base[0])[0][0])
    b1 = int(np.nonzero(ta.year <= base[1])[0][-1]) + 1
    p = pdoy(x[b0:b1], ta.subset(slice(b0, b1)), 5, 90.0, device=dev)
    # the index-level spelling: tx90p(..., bootstrap=True) reads the base period and window from the percentile attrs
    got = xi.tx90p(x, p, ta, freq=freq, device=dev, bootstrap=True, mask_missing=False)
    exp = oboot.bootstrap_exceedance(x, ot, base, freq, ">", 5, 90.0)
    assert got.shape == exp.shape
    np.testing.assert_array_equal(got, exp)
    # in-base years see fewer exceedances without the bootstrap (Zhang 2005; reference tests/test_bootstrapping.py:24-75)
    plain = xi.tx90p(x, p, ta, freq=freq, device=dev, mask_missing=False)
    seg, starts = ta.segments(freq)
    inb = np.array([base[0] <= (y if (freq == "YS" or m != 12) else y + 1) <= base[1] for y, m in starts])
    assert got[inb].sum() >= plain[inb].sum()
    # out-of-base periods are the plain index against the supplied percentile (tests/test_bootstrapping.py:73-75)
    outb = np.array([not (base[0] <= y <= base[1]) and not (base[0] <= (y + 1 if m == 12 else y) <= base[1]) for y, m in starts])
    np.testing.assert_array_equal(got[outb], plain[outb])
    # with the Indicator-level MissingAny mask the bootstrap changes values, never which periods are missing
    masked = xi.tx90p(x, p, ta, freq=freq, device=dev, bootstrap=True)
    np.testing.assert_array_equal(masked, oidx.apply_missing(exp, x, ot, freq))
    with pytest.raises(KeyError):  # percentile from the whole series: nothing to bootstrap (bootstrapping.py:157-162)
        xi.tx90p(x, pdoy(x, ta, 5, 90.0, device=dev), ta, freq=freq, device=dev, bootstrap=True)
    with pytest.raises(KeyError):  # no overlap at all (:163-168)
        far = ta.subset(slice(0, 400))
        pfar = pdoy(x[:400], far, 5, 90.0, device=dev)
        pfar.attrs["climatology_bounds"] = ["1950-01-01", "1951-12-31"]
        xi.tx90p(x, pfar, ta, freq=freq, device=dev, bootstrap=True)


@pytest.mark.parametrize("name,per,freq", [("warm_spell_duration_index", 90.0, "MS"), ("cold_spell_duration_index", 10.0, "MS"),
                                           ("warm_spell_duration_index", 90.0, "YS"), ("tn10p", 10.0, "YS-JUL"),
                                           ("tg90p", 90.0, "QS-APR")])
def test_bootstrap_is_generic_over_the_decorated_indices(dev, rng, name, per, freq):
    """core/bootstrapping.py:81-211 bootstraps ANY decorated index; the reference decorates the t*10p / t*90p family,
    warm / cold_spell_duration_index and the two precipitation indices (indices/_multivariate.py:68 ... 1718) and its
    test parametrises WSDI / CSDI with "MS" (tests/test_bootstrapping.py:24-41).  Oracle: replicas materialised like
    `build_bootstrap_year_da`, the index evaluated on each."""
    from oracle import bootstrapping as oboot
    from xclim_amd.calendar import percentile_doy as pdoy

    T = 365 * 5
    x = _temp(rng, T, (2, 5))
    # a strongly autocorrelated signal so that spells of 3+ days above the percentile exist
    x = (x[:: 1] * 0 + np.repeat(_temp(rng, T // 5 + 1, (2, 5)), 5, axis=0)[:T]).astype(np.float32)
    ta, ot = _axes("2000-01-01", T, "noleap")
    nb = 365 * 3
    p = pdoy(x[:nb], ta.subset(slice(0, nb)), 5, per, device=dev)
    f = getattr(xi, name)
    kw = dict(window=3) if "spell" in name else {}
    got = f(x, p, ta, freq=freq, device=dev, bootstrap=True, mask_missing=False, **kw)
    if "spell" in name:
        of = oidx.warm_spell_duration_index if name.startswith("warm") else oidx.cold_spell_duration_index
        index_fn = lambda xx, pp, dd, t: of(xx, pp, dd, t, 3, freq)  # noqa: E731
        op = ">" if name.startswith("warm") else "<"
    else:
        index_fn, op = None, (">" if name.endswith("90p") else "<")
    exp = oboot.bootstrap_exceedance(x, ot, (2000, 2002), freq, op, 5, per, index_fn=index_fn)
    np.testing.assert_allclose(got, exp, rtol=0, atol=1e-12)
    assert np.nansum(exp) > 0
    plain = f(x, p, ta, freq=freq, device=dev, mask_missing=False, **kw)
    assert not np.array_equal(got, plain)  # the in-base years did change


def test_bootstrap_uses_the_supplied_percentile_outside_the_overlap(dev, rng):
    """bootstrapping.py:196-199: years of `da` outside its overlap with the reference period are evaluated against the
    percentile the CALLER passed — which may come from a longer period than `da` covers (it used to be
    recomputed from the overlap)."""
    from oracle import bootstrapping as oboot
    from xclim_amd.calendar import percentile_doy as pdoy

    T = 365 * 8
    x = _temp(rng, T, (6,))
    ta, ot = _axes("2000-01-01", T, "noleap")
    p = pdoy(x[: 365 * 5], ta.subset(slice(0, 365 * 5)), 5, 90.0, device=dev)       # reference period 2000-2004
    sl = slice(365 * 3, T)                                                            # studied series 2003-2007
    got = xi.tx90p(x[sl], p, ta.subset(sl), freq="YS", device=dev, bootstrap=True, mask_missing=False)
    pe, de = ocal.percentile_doy(x[: 365 * 5], ot.isel(np.arange(365 * 5)), 5, 90.0)
    exp = oboot.bootstrap_exceedance(x[sl], ot.isel(np.arange(365 * 3, T)), (2000, 2004), "YS", ">", 5, 90.0, per_out=(pe, de))
    np.testing.assert_array_equal(got, exp)


def test_bivariate_and_thresholded_generic(dev, rng):
    T = 800
    tn = _temp(rng, T, (5, 4), nan_frac=0.01) - 15
    tx = tn + np.abs(rng.normal(6, 2, tn.shape)).astype(np.float32)
    ta, ot = _axes("2000-05-01", T)
    for freq in ("YS", "MS"):
        np.testing.assert_array_equal(xgen.count_level_crossings(tn, tx, 273.15, ta, freq, device=dev),
                                      ogen.count_level_crossings(tn, tx, 273.15, ot, freq))
        for red in ("all", "any"):
            got = xgen.bivariate_count_occurrences(data_var1=tn, data_var2=tx, threshold_var1=270.0, threshold_var2=285.0, time=ta,
                                                   freq=freq, op_var1="<", op_var2=">", var_reducer=red, device=dev)
            np.testing.assert_array_equal(got, ogen.bivariate_count_occurrences(tn, tx, 270.0, 285.0, ot, freq, "<", ">", red))
        for red in ("sum", "mean", "min", "max"):
            got = xgen.thresholded_statistics(tx, ">", 285.0, red, ta, freq, device=dev)
            np.testing.assert_allclose(got, ogen.thresholded_statistics(tx, ">", 285.0, red, ot, freq), rtol=1e-6, equal_nan=True)
        for op in (">", "<="):
            np.testing.assert_allclose(xgen.temperature_sum(tx, op, 283.0, ta, freq, device=dev),
                                       ogen.temperature_sum(tx, op, 283.0, ot, freq), rtol=1e-6, atol=1e-4)
            np.testing.assert_allclose(xgen.cumulative_difference(tx, 283.0, op, ta, freq, device=dev),
                                       ogen.cumulative_difference(tx, 283.0, op, ot, freq), rtol=1e-6, atol=1e-4)
    with pytest.raises(NotImplementedError):
        xgen.cumulative_difference(tx, 283.0, "==", ta, "YS", device=dev)
    with pytest.raises(ValueError):
        xgen.bivariate_count_occurrences(data_var1=tn, data_var2=tx, threshold_var1=0, threshold_var2=0, time=ta, freq="YS",
                                         op_var1="<", op_var2=">", var_reducer="some", device=dev)


def test_occurrence_and_doy_extremes(dev, rng):
    """first/last_occurrence, first_day_threshold_reached (gen:1108-1201, 1555-1608), doymax/doymin (gen:177-221)."""
    T = 365 * 2 + 150
    x = _temp(rng, T, (4, 5)) - 273.15
    x[:, 0, 0] = 30.0  # constant: all-True condition (first_run quirk) and std == 0 (doymax -> NaN)
    ta, ot = _axes("2001-01-01", T)

    def omap(fn, cond, *a):
        res = orl.map_groups_fn(fn, cond, ot, "YS", *a)
        out = np.full(res.shape, np.nan)
        for p, (_, idx) in enumerate(__import__("oracle.timeutil", fromlist=["groups"]).groups(ot, "YS")):
            ok = ~np.isnan(res[p])
            out[p][ok] = ot.doy[idx[0] + res[p][ok].astype(int)]
        return out

    cond = x > 20.0
    np.testing.assert_array_equal(xgen.first_occurrence(x, 20.0, ">", ta, "YS", device=dev),
                                  omap(lambda g, t: orl.first_run(g, 1), cond))
    np.testing.assert_array_equal(xgen.last_occurrence(x, 20.0, ">", ta, "YS", device=dev),
                                  omap(lambda g, t: orl.last_run(g, 1), cond))
    got = xgen.first_day_threshold_reached(x, threshold=15.0, op=">", after_date="03-01", time=ta, window=3, freq="YS", device=dev)
    np.testing.assert_array_equal(got, omap(lambda g, w, d, t: orl.first_run_after_date(g, w, d, t), x > 15.0, 3, "03-01"))
    dm = xgen.doymax(x, ta, "YS", device=dev)
    assert np.isnan(dm[:, 0, 0]).all()
    seg, _ = ta.segments("YS")
    exp = np.stack([ta.doy[seg[p] + np.argmax(x[seg[p]:seg[p + 1]], axis=0)] for p in range(len(seg) - 1)]).astype(float)
    exp[:, 0, 0] = np.nan
    np.testing.assert_array_equal(dm, exp)


@pytest.mark.parametrize("calendar", ["standard", "noleap"])
def test_climatological_mean_doy(dev, rng, calendar):
    from xclim_amd.calendar import climatological_mean_doy

    T = 365 * 4 + (1 if calendar == "standard" else 0)
    x = _temp(rng, T, (3, 5), nan_frac=0.01)
    ta, ot = _axes("2000-01-01", T, calendar)
    m, s, doys = climatological_mean_doy(x, ta, window=5, device=dev)
    em, es, edoys = ocal.climatological_mean_doy(x, ot, 5)
    np.testing.assert_array_equal(doys, edoys)
    np.testing.assert_allclose(m, em, rtol=1e-6)
    np.testing.assert_allclose(s, es, rtol=2e-6, atol=1e-6)


@pytest.mark.parametrize("k", range(7))
def test_calc_perc_on_reference_golden_vectors(dev, k):
    """The apply_ufunc callee (utl:279-323) through the HIP path against the reference's own outputs."""
    x = GOLD[f"q_in_{k}"]
    pers = list(GOLD["q_pers"])
    for typ, ab in (("t7", (1.0, 1.0)), ("t8", (1 / 3, 1 / 3))):
        got = xutils.calc_perc(x, pers, *ab, device=dev)
        np.testing.assert_allclose(got, GOLD[f"q_{typ}_{k}"], rtol=1e-12, atol=0, equal_nan=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        np.testing.assert_allclose(xutils.calc_perc(x, pers, device=dev), oq.calc_perc(x, pers), rtol=1e-12, equal_nan=True)


@pytest.mark.parametrize("kind,interp", [("+", "nearest"), ("*", "linear")])
def test_eqm_object_api(dev, rng, kind, interp):
    T, shape = 1095, (6, 7)
    ref = _temp(rng, T, shape)
    hist = (_temp(rng, T, shape) * 1.01 + 1.5).astype(np.float32)
    sim = (_temp(rng, T, shape, nan_frac=0.01) + 2).astype(np.float32)
    eqm = xsdba.EmpiricalQuantileMapping.train(ref, hist, nquantiles=20, kind=kind, group="time", device=dev)
    eaf, ehq = osdba.eqm_train(ref.reshape(T, -1), hist.reshape(T, -1), 20, kind)
    np.testing.assert_allclose(eqm.hist_q.reshape(20, -1), ehq, rtol=1e-6)
    np.testing.assert_allclose(eqm.af.reshape(20, -1), eaf, rtol=1e-5, atol=1e-5 if kind == "+" else 0)
    scen = eqm.adjust(sim, interp=interp, extrapolation="constant")
    exp = osdba.eqm_adjust(sim.reshape(T, -1), eqm.af.reshape(20, -1), eqm.hist_q.reshape(20, -1), kind, interp, "constant")
    np.testing.assert_allclose(scen.reshape(T, -1), exp, rtol=1e-6, equal_nan=True)
    assert eqm.adj_params["kind"] == kind
    np.testing.assert_allclose(xsdba.quantile(ref, eqm.quantiles, device=dev).reshape(20, -1), osdba.quantile(ref.reshape(T, -1), eqm.quantiles), rtol=1e-6)
    with pytest.raises(NotImplementedError):
        xsdba.EmpiricalQuantileMapping.train(ref, hist, group="time.week", device=dev)


@pytest.mark.parametrize("kind", ["+", "*"])
@pytest.mark.parametrize("interp", ["linear", "nearest"])
def test_eqm_reference_testqm(dev, kind, interp):
    """The reference's only numeric test at the sdba boundary (tests/test_xsdba.py:113-155, TestQM.test_quantiles), its
    own data path and tolerances through the HIP kernels: hist = sim ~ U(10, 11), ref ~ N(12, 1) drawn from the SAME
    10000 uniform numbers, nquantiles = 50, group "time"; ``af[2:-2]`` equals the correction of the theoretical
    quantiles to 1 decimal and ``adjust(sim)`` reproduces ``ref`` to 1 decimal away from the extremes — for both kinds
    (the reference adjusts with interp="linear"; "nearest", a step function, is held to half a node spacing)."""
    from scipy.stats import norm, uniform

    u = np.random.default_rng(0).random(10000)
    xd, yd = uniform(loc=10, scale=1), norm(loc=12, scale=1)
    x, y = xd.ppf(u).astype(np.float32)[:, None], yd.ppf(u).astype(np.float32)[:, None]
    qm = xsdba.EmpiricalQuantileMapping.train(y, x, kind=kind, group="time", nquantiles=50, device=dev)
    q = qm.quantiles
    expected = yd.ppf(q) - xd.ppf(q) if kind == "+" else yd.ppf(q) / xd.ppf(q)      # sdba.utils.get_correction(x_q, y_q, kind)
    np.testing.assert_array_almost_equal(qm.af[2:-2, 0], expected[2:-2], 1)
    p = qm.adjust(x, interp=interp)
    middle = (u > 1e-2) & (u < 0.99)
    if interp == "linear":
        np.testing.assert_array_almost_equal(p[middle, 0], y[middle, 0], 1)
    else:  # a step function of 50 nodes: half a node spacing of the correction in the tails of the normal
        assert np.abs(p[middle, 0] - y[middle, 0]).max() < 0.35


def test_range_reductions_and_daily_events(dev, rng):
    """diurnal / interday / extreme temperature range (gen:1076-1105, 1360-1414), compare and get_daily_events."""
    T = 800
    tn = _temp(rng, T, (5, 4), nan_frac=0.02)
    tx = tn + np.abs(rng.normal(6, 2, tn.shape)).astype(np.float32)
    tx[rng.random(tx.shape) < 0.02] = np.nan
    tn[:40, 0, 0] = np.nan  # an all-NaN month
    ta, ot = _axes("2000-05-01", T)
    for freq in ("YS", "MS", "QS-DEC"):
        for red in ("max", "min", "mean", "sum"):
            got = xgen.diurnal_temperature_range(tn, tx, red, ta, freq, device=dev)
            ref = ogen.diurnal_temperature_range(tn, tx, red, ot, freq)
            np.testing.assert_allclose(got, ref, rtol=1e-6, atol=1e-5 if red == "sum" else 0, equal_nan=True)
        np.testing.assert_allclose(xgen.interday_diurnal_temperature_range(tn, tx, ta, freq, device=dev),
                                   ogen.interday_diurnal_temperature_range(tn, tx, ot, freq), rtol=1e-6, equal_nan=True)
        np.testing.assert_allclose(xgen.extreme_temperature_range(tn, tx, ta, freq, device=dev),
                                   ogen.extreme_temperature_range(tn, tx, ot, freq), rtol=1e-6, equal_nan=True)
    out, valid = xgen.diurnal_temperature_range(tn, tx, "mean", ta, "MS", device=dev, with_valid=True)
    seg, _ = ta.segments("MS")
    both = (~np.isnan(tn) & ~np.isnan(tx)).astype(np.int32)
    np.testing.assert_array_equal(valid, np.stack([both[a:b].sum(axis=0) for a, b in zip(seg[:-1], seg[1:])]))
    for op in (">", "<=", "==", "!="):
        for thr in (285.0, np.float64(285.1)):
            np.testing.assert_array_equal(xgen.compare(tx, op, thr, device=dev), ogen.compare(tx, op, thr))
            np.testing.assert_array_equal(xgen.get_daily_events(tx, thr, op, device=dev), ogen.get_daily_events(tx, thr, op))
    np.testing.assert_array_equal(xgen.compare(tx, ">", tn + np.float32(6.0), device=dev), ogen.compare(tx, ">", tn + np.float32(6.0)))
    with pytest.raises(ValueError):
        xgen.compare(tx, "<", 1.0, constrain=(">", ">="), device=dev)
    with pytest.raises(ValueError):
        xgen.diurnal_temperature_range(tn, tx, "std", ta, "YS", device=dev)


@pytest.mark.gpu
@pytest.mark.parametrize("calendar,T", [("standard", 1461), ("noleap", 1095)])
def test_resample_doy_and_within_bnds_doy(dev, rng, calendar, T):
    """cal:763-790 / 934-954: per-doy tables broadcast onto the time axis."""
    from xclim_amd.calendar import resample_doy, within_bnds_doy

    x = _temp(rng, T, (4, 5), nan_frac=0.01)
    ta, ot = _axes("2000-01-01", T, calendar)
    p = percentile_doy(x, ta, window=5, per=[10.0, 90.0], device=dev)
    exp, doys = ocal.percentile_doy(x, ot, 5, [10.0, 90.0])
    # target axis in another span (incl. other leap years)
    T2 = 900
    y = _temp(rng, T2, (4, 5), nan_frac=0.01)
    ta2, ot2 = _axes("2003-01-01", T2, calendar)  # (the oracle noleap axis starts on Jan 1)
    lo = resample_doy(p.sel(10.0), ta2, device=dev)
    hi = resample_doy(p.sel(90.0), ta2, device=dev)
    np.testing.assert_allclose(lo, ocal.resample_doy(exp[..., 0], doys, ot2), rtol=1e-12, equal_nan=True)
    np.testing.assert_allclose(hi, ocal.resample_doy(exp[..., 1], doys, ot2), rtol=1e-12, equal_nan=True)
    got = within_bnds_doy(y, low=p.sel(10.0), high=p.sel(90.0), time=ta2, device=dev)
    with np.errstate(invalid="ignore"):
        ref = (ocal.resample_doy(exp[..., 0], doys, ot2) < y) * (y < ocal.resample_doy(exp[..., 1], doys, ot2))
    np.testing.assert_array_equal(got, ref)
    assert got.any() and not got.all()


@pytest.mark.gpu
@pytest.mark.parametrize("calendar,T,freq,window,per,op", [("noleap", 365, "YS", 5, 90.0, ">"), ("noleap", 365, "MS", 5, 10.0, "<"),
                                                           ("noleap", 365, "QS-DEC", 3, 50.0, ">="), ("standard", 365, "MS", 7, 90.0, ">"),
                                                           ("noleap", 200, "MS", 5, 75.0, ">"), ("noleap", 1095, "YS", 5, 90.0, ">"),
                                                           ("standard", 366, "YS", 5, 90.0, ">"),
                                                           ("noleap", 365 * 12, "YS", 5, 95.0, ">"), ("noleap", 365 * 9, "MS", 7, 3.0, "<="),
                                                           ("360_day", 360 * 8, "QS-DEC", 5, 90.0, ">=")])
def test_percentile_exceedance_fused(dev, rng, calendar, T, freq, window, per, op):
    """The fused percentile_doy + threshold_count kernels (one year: sliding window; several years on a calendar without
    gaps: register top-16 + per-year counters) must reproduce the two-step chain bit for bit, and the wrapper falls back
    to the chain where they do not apply (366-day years, central percentiles on several years)."""
    x = _temp(rng, T, (7, 9), nan_frac=0.01)
    x[10:13, 0, 0] = np.nan
    start = "2001-01-01" if T != 366 else "2000-01-01"
    ta, ot = _axes(start, T, calendar)
    got = xi.percentile_exceedance(x, ta, freq, op, window, per, device=dev)
    p = percentile_doy(x, ta, window=window, per=per, device=dev)
    f = xi.tx90p if op in (">", ">=") else xi.tx10p
    ref = f(x, p, ta, freq, op, device=dev)
    np.testing.assert_array_equal(got, ref)
    exp, doys = ocal.percentile_doy(x, ot, window, per)
    raw = xi.percentile_exceedance(x, ta, freq, op, window, per, device=dev, mask_missing=False)
    of = oidx.tx90p if op in (">", ">=") else oidx.tx10p
    np.testing.assert_array_equal(raw, of(x, exp[..., 0], doys, ot, freq, op))


@pytest.mark.gpu
@pytest.mark.parametrize("calendar,T", [("standard", 1461), ("noleap", 1095)])
@pytest.mark.parametrize("before", [True, False])
def test_percentile_spell_indices(dev, rng, calendar, T, before):
    """warm / cold spell duration index (indices/_multivariate.py:66-152, 1693-1793): per-doy percentile threshold ->
    windowed_run_count; the first `next` row of SURVEY.md section 8f."""
    x = _temp(rng, T, (5, 6), nan_frac=0.004)
    # slow anomalies so that multi-day spells above / below the percentiles exist
    x += np.repeat(rng.normal(0, 3.0, (T // 7 + 1, 5, 6)), 7, axis=0)[:T].astype(np.float32)
    ta, ot = _axes("2000-01-01", T, calendar)
    p = percentile_doy(x, ta, window=5, per=[25.0, 75.0], device=dev)
    exp, doys = ocal.percentile_doy(x, ot, 5, [25.0, 75.0])
    for freq in ("YS", "MS"):
        for window in (2, 6):
            got = xi.warm_spell_duration_index(x, p.sel(75.0), ta, window, freq, before, device=dev, mask_missing=False)
            ref = oidx.warm_spell_duration_index(x, exp[..., 1], doys, ot, window, freq, before)
            np.testing.assert_array_equal(got, ref)
            got = xi.cold_spell_duration_index(x, p.sel(25.0), ta, window, freq, before, device=dev, mask_missing=False)
            ref = oidx.cold_spell_duration_index(x, exp[..., 0], doys, ot, window, freq, before)
            np.testing.assert_array_equal(got, ref)
    assert xi.warm_spell_duration_index(x, p.sel(75.0), ta, 2, "YS", device=dev, mask_missing=False).sum() > 0
    masked = xi.warm_spell_duration_index(x, p.sel(75.0), ta, 6, "MS", device=dev)
    nan_month = np.isnan(masked)
    seg, _ = ta.segments("MS")
    has_nan = np.stack([np.isnan(x[a:b]).any(axis=0) for a, b in zip(seg[:-1], seg[1:])])
    np.testing.assert_array_equal(nan_month, has_nan)
    with pytest.raises(ValueError):
        xi.warm_spell_duration_index(x, p.sel(75.0), ta, op="<", device=dev)


@pytest.mark.parametrize("group,window,nyears", [("time.month", 1, 4), ("time.dayofyear", 1, 3), ("time.dayofyear", 31, 3),
                                                 ("time.dayofyear", 7, 2), ("time.season", 1, 3)])
@pytest.mark.parametrize("kind,interp", [("+", "nearest"), ("*", "nearest"), ("+", "linear")])
def test_eqm_with_sub_groupings(dev, rng, group, window, nyears, kind, interp):
    """EmpiricalQuantileMapping with xsdba's Grouper("time.month") / Grouper("time.dayofyear", window) (SURVEY 8f rank 4,
    first slice).  PARITY UNPINNED (xsdba is not available): the oracle is the specified restatement in oracle/sdba.py —
    group samples = centred window around every step of the group, Hyndman-Fan type 7 nodes, every step adjusted with
    the factors of its own group — plus the analytic property a grouped mapping must have (below)."""
    T = 365 * nyears
    ta, ot = _axes("2001-01-01", T, "noleap")
    shape = (3, 5)
    ref = _temp(rng, T, shape, nan_frac=0.002)
    hist = (_temp(rng, T, shape) * (1.1 if kind == "*" else 1.0) + 1.5).astype(np.float32)
    sim = (_temp(rng, T + 40, shape, nan_frac=0.002) + 2.0).astype(np.float32)[40:]
    prop = group.split(".")[1]
    eqm = xsdba.EmpiricalQuantileMapping.train(ref, hist, nquantiles=15, kind=kind, group=group, window=window, time=ta, device=dev)
    oaf, ohq, labels = osdba.eqm_train_grouped(ref, hist, ot, prop, window, 15, kind)
    assert eqm.af.shape == oaf.shape == (len(labels), 15) + shape
    np.testing.assert_array_equal(eqm.group_labels, labels)
    np.testing.assert_allclose(eqm.hist_q, ohq, rtol=1e-6, equal_nan=True)
    np.testing.assert_allclose(eqm.af, oaf, rtol=1e-6, atol=1e-5, equal_nan=True)
    if interp != "nearest":  # xsdba interpolates over the (quantile, group) PLANE there (round 5: xh_plane_linear)
        if prop == "season":  # upstream's season coordinate is not restated: refused, not approximated
            with pytest.raises(NotImplementedError, match="plane"):
                eqm.adjust(sim, interp=interp, extrapolation="constant", time=ta)
            return
        scen = eqm.adjust(sim, interp=interp, extrapolation="constant", time=ta)
        exp = osdba.eqm_adjust_grouped(sim, ot, prop, labels, eqm.af, eqm.hist_q, kind, interp, "constant", mode="griddata")
        np.testing.assert_allclose(scen, exp, rtol=1e-6, atol=1e-6, equal_nan=True)
        with pytest.raises(NotImplementedError, match="Clough-Tocher"):
            eqm.adjust(sim, interp="cubic", time=ta)
        with pytest.raises(NotImplementedError, match="convex hull"):
            eqm.adjust(sim, interp=interp, extrapolation="nan", time=ta)
        return
    scen = eqm.adjust(sim, interp=interp, extrapolation="constant", time=ta, grouped_nearest="group")
    exp = osdba.eqm_adjust_grouped(sim, ot, prop, labels, eqm.af, eqm.hist_q, kind, interp, "constant")
    np.testing.assert_allclose(scen, exp, rtol=1e-6, equal_nan=True)
    if prop != "season":  # the default since round 4: xsdba's nearest node in the (hist_q, group) plane (scipy griddata in the oracle)
        for extrap in ("constant", "nan"):
            scen2 = eqm.adjust(sim, interp=interp, extrapolation=extrap, time=ta)
            exp2 = osdba.eqm_adjust_grouped(sim, ot, prop, labels, eqm.af, eqm.hist_q, kind, interp, extrap, mode="griddata")
            bad = ~np.isclose(scen2, exp2, rtol=1e-6, atol=0, equal_nan=True)
            if bad.any() and window > 1:   # (exact ties between neighbouring windowed groups, on some draws of the inputs)
                bad = _accept_plane_nearest_ties(bad, scen2, sim, osdba.group_values(ot, prop), eqm.hist_q, eqm.af, kind)
            assert not bad.any(), f"{extrap}: {int(bad.sum())} mismatches"
    else:
        np.testing.assert_array_equal(eqm.adjust(sim, time=ta), scen)   # seasons keep the own-group rule
    assert "Grouper" in str(eqm.adj_params["group"]) or eqm.adj_params["group"] == group
    with pytest.raises(ValueError, match="needs time"):
        eqm.adjust(sim)
    with pytest.raises(ValueError, match="needs time"):
        xsdba.EmpiricalQuantileMapping.train(ref, hist, group=group, window=window, device=dev)


def test_grouped_nearest_in_the_value_group_plane(dev, rng):
    """xsdba's grouped "nearest" = scipy griddata in the (hist_q, group coordinate) plane: with precipitation-like values
    (nodes tens of mm/day apart in the upper tail) a node of a NEIGHBOURING month is often nearer than the own month's —
    round 3 always took the own month's node (its documented deviation).  GPU == the oracle's restatement with the real
    griddata, for both kinds and extrapolations, incl. the cyclic December <-> January neighbourhood and NaN nodes; the
    two rules do differ on this field."""
    T = 365 * 5
    ta, ot = _axes("2001-01-01", T, "noleap")
    shape = (2, 4)
    seas = 1.0 + 0.6 * np.sin(2 * np.pi * (np.arange(T) - 30) / 365)[:, None, None]

    def pr(scale):
        x = rng.gamma(0.6, scale, (T,) + shape) * seas
        x[rng.random(x.shape) < 0.5] = 0.0
        return x.astype(np.float32)

    ref, hist, sim = pr(9.0), pr(7.0), pr(8.0)
    sim[rng.random(sim.shape) < 0.01] = np.nan
    eqm = xsdba.EmpiricalQuantileMapping.train(ref, hist, nquantiles=20, kind="*", group="time.month", time=ta, device=dev)
    labels = eqm.group_labels
    a_plane = eqm.adjust(sim, time=ta)
    a_group = eqm.adjust(sim, time=ta, grouped_nearest="group")
    exp = osdba.eqm_adjust_grouped(sim, ot, "month", labels, eqm.af, eqm.hist_q, "*", "nearest", "constant", mode="griddata")
    np.testing.assert_allclose(a_plane, exp, rtol=1e-6, equal_nan=True)
    differ = ~np.isclose(a_plane, a_group, rtol=1e-6, equal_nan=True)
    assert differ.mean() > 0.01                                       # the deviation of rounds 2-3 was not a corner case here
    jan, dec = ta.month == 1, ta.month == 12
    assert differ[jan].any() and differ[dec].any()                    # (the cyclic copies at coordinates 0 and 13 take part)
    np.testing.assert_allclose(eqm.adjust(sim, time=ta, extrapolation="nan"),
                               osdba.eqm_adjust_grouped(sim, ot, "month", labels, eqm.af, eqm.hist_q, "*", "nearest", "nan", mode="griddata"),
                               rtol=1e-6, equal_nan=True)


def test_grouped_eqm_removes_a_seasonal_bias(dev, rng):
    """Analytic check in the spirit of the reference's only numeric sdba test (tests/test_xsdba.py:113-155): a bias that
    depends on the month is removed by group="time.month" (every month's quantiles shift by that month's bias, so the
    additive factors are minus the bias at every node) and NOT by group="time"."""
    T = 365 * 6
    ta, _ = _axes("2001-01-01", T, "noleap")
    ref = _temp(rng, T, (4,))
    bias = np.linspace(-3.0, 3.0, 12).astype(np.float32)[ta.month - 1][:, None]
    hist = (ref + bias).astype(np.float32)
    by_month = xsdba.EmpiricalQuantileMapping.train(ref, hist, nquantiles=20, kind="+", group="time.month", time=ta, device=dev)
    np.testing.assert_allclose(by_month.af, np.broadcast_to(-np.linspace(-3.0, 3.0, 12)[:, None, None], by_month.af.shape), atol=2e-4)
    # (own-group nearest: with xsdba's nearest node in the (value, group) plane a neighbouring month's node wins at a few
    # steps in the tails and the removal is no longer exact)
    np.testing.assert_allclose(by_month.adjust(hist, time=ta, grouped_nearest="group"), ref, atol=5e-4)
    assert (np.abs(by_month.adjust(hist, time=ta) - ref) > 5e-4).mean() < 0.01
    flat = xsdba.EmpiricalQuantileMapping.train(ref, hist, nquantiles=20, kind="+", device=dev)
    assert np.abs(flat.adjust(hist) - ref).max() > 1.0
    with pytest.raises(NotImplementedError):
        xsdba.Grouper("time.week")
    assert list(xsdba.Grouper("time.season").labels(ta)) == ["DJF", "JJA", "MAM", "SON"]   # sorted like xarray's groupby
    with pytest.raises(ValueError):
        xsdba.Grouper("time.month", window=4)
    with pytest.raises(ValueError):
        xsdba.Grouper("time", window=3)


@pytest.mark.gpu
@pytest.mark.parametrize("R", [3, 11, 40, 128])
def test_weighted_ensemble_percentiles(dev, rng, R):
    """ensembles/_base.py:346-356: with `weights` the reference calls xarray's weighted quantile (PARITY UNPINNED: xarray is
    not available; oracle and kernel restate its published algorithm).  Pinned analytically: equal weights reproduce the
    unweighted "linear" (type 7) percentiles; a zero weight removes the member; NaN members are skipped; only "linear"."""
    from oracle import ensembles as oens
    from xclim_amd import ensembles as xens

    ens = rng.normal(10, 3, (R, 5, 9)).astype(np.float32)
    ens[rng.random(ens.shape) < 0.1] = np.nan
    ens[:, 0, 0] = np.nan
    vals = [5, 10, 50, 90, 99]
    w = rng.random(R) + 0.1
    got = xens.ensemble_percentiles(ens, vals, weights=w, device=dev)
    np.testing.assert_allclose(got, oens.ensemble_percentiles(ens, vals, weights=w), rtol=1e-12, atol=1e-12, equal_nan=True)
    assert np.isnan(got[0, 0]).all()
    # equal weights == the unweighted estimator (both the oracle's calc_perc, pinned to the reference, and the HIP path)
    eq = xens.ensemble_percentiles(ens, vals, weights=np.full(R, 2.5), device=dev)
    np.testing.assert_allclose(eq, oens.ensemble_percentiles(ens, vals), rtol=1e-6, equal_nan=True)
    # a zero weight drops the member
    w0 = w.copy()
    w0[0] = 0.0
    np.testing.assert_allclose(xens.ensemble_percentiles(ens, vals, weights=w0, device=dev),
                               xens.ensemble_percentiles(ens[1:], vals, weights=w[1:], device=dev), rtol=1e-12, equal_nan=True)
    mm = xens.ensemble_percentiles(ens, vals, weights=w, min_members=R, device=dev)
    assert np.isnan(mm[np.isnan(ens).any(axis=0)]).all()
    with pytest.raises(ValueError, match="Only the 'linear' method"):
        xens.ensemble_percentiles(ens, vals, weights=w, method="hazen", device=dev)
    if R == 128:
        with pytest.raises(Exception):
            xens.ensemble_percentiles(np.zeros((129, 4), np.float32), vals, weights=np.ones(129), device=dev)


@pytest.mark.parametrize("method", ["linear", "median_unbiased", "weibull", "hazen"])
@pytest.mark.parametrize("R", [5, 21, 60])
def test_ensemble_percentiles(dev, rng, method, R):
    """ensembles/_base.py:213-372 (second caller of calc_perc): percentiles over the realization axis, min_members."""
    from oracle import ensembles as oens
    from xclim_amd import ensembles as xens

    ens = rng.normal(10.0, 3.0, (R, 24, 7, 5)).astype(np.float32)
    ens[rng.random(ens.shape) < 0.1] = np.nan
    ens[:, 0, 0, 0] = np.nan
    ens[: R - 2, 1, 1, 1] = np.nan
    for mm in (1, None, max(1, R // 2)):
        got = xens.ensemble_percentiles(ens, [10, 50, 90], min_members=mm, method=method, device=dev)
        ref = oens.ensemble_percentiles(ens, [10, 50, 90], min_members=mm, method=method)
        assert got.shape == ref.shape == (24, 7, 5, 3)
        np.testing.assert_allclose(got, ref, rtol=1e-12, equal_nan=True)
    # reference known answer (tests/test_ensembles.py style): percentiles of 0..R-1 with the linear method
    lin = np.arange(R, dtype=np.float32)[:, None]
    got = xens.ensemble_percentiles(lin, [0, 50, 100], device=dev)
    np.testing.assert_allclose(got[0], [0, (R - 1) / 2, R - 1], rtol=1e-12)
    if method != "linear":  # _base.py:347-348
        with pytest.raises(ValueError, match="Only the 'linear' method"):
            xens.ensemble_percentiles(ens, weights=np.ones(R), method=method, device=dev)


@pytest.mark.gpu
@pytest.mark.parametrize("calendar,T", [("standard", 1461), ("noleap", 800)])
def test_index_level_callers(dev, rng, calendar, T):
    """Index functions of indices/_threshold.py, _simple.py, _multivariate.py as compositions over the hot path."""
    tas = _temp(rng, T, (4, 5), nan_frac=0.003)
    tas += np.repeat(rng.normal(0, 4.0, (T // 5 + 1, 4, 5)), 5, axis=0)[:T].astype(np.float32)
    tn = tas - np.abs(rng.normal(4, 1.5, tas.shape)).astype(np.float32)
    tx = tas + np.abs(rng.normal(4, 1.5, tas.shape)).astype(np.float32)
    pr = (rng.gamma(0.5, 4.0, tas.shape) * (rng.random(tas.shape) < 0.5)).astype(np.float32)
    ta, ot = _axes("2000-01-01", T, calendar)
    kw = dict(device=dev, mask_missing=False)
    for freq in ("YS", "MS"):
        np.testing.assert_array_equal(xi.tx_days_above(tx, 295.0, ta, freq, **kw), oidx.count_days(tx, ">", 295.0, ot, freq))
        np.testing.assert_array_equal(xi.tn_days_below(tn, 270.0, ta, freq, "<=", **kw), oidx.count_days(tn, "<=", 270.0, ot, freq))
        np.testing.assert_array_equal(xi.ice_days(tx, 273.15, ta, freq, **kw), oidx.count_days(tx, "<", 273.15, ot, freq))
        np.testing.assert_array_equal(xi.dry_days(pr, 0.2, ta, freq, **kw), oidx.count_days(pr, "<", 0.2, ot, freq))
        np.testing.assert_array_equal(xi.wetdays(pr, 1.0, ta, freq, **kw), oidx.count_days(pr, ">=", 1.0, ot, freq))
        for before in (True, False):
            np.testing.assert_array_equal(xi.hot_spell_frequency(tx, 292.0, ta, 3, freq, ">", before, **kw),
                                          oidx.run_index(tx, ">", 292.0, "events", 3, ot, freq, before))
            np.testing.assert_array_equal(xi.hot_spell_total_length(tx, 292.0, ta, 3, freq, ">", before, **kw),
                                          oidx.run_index(tx, ">", 292.0, "count", 3, ot, freq, before))
            np.testing.assert_array_equal(xi.hot_spell_max_length(tx, 292.0, ta, 4, freq, ">", before, **kw),
                                          oidx.longest_run_index(tx, ">", 292.0, 4, ot, freq, before))
            np.testing.assert_array_equal(xi.cold_spell_days(tas, 275.0, ta, 5, freq, "<", before, **kw),
                                          oidx.run_index(tas, "<", 275.0, "count", 5, ot, freq, before))
            np.testing.assert_array_equal(xi.cold_spell_frequency(tas, 275.0, ta, 5, freq, "<", before, **kw),
                                          oidx.run_index(tas, "<", 275.0, "events", 5, ot, freq, before))
            np.testing.assert_array_equal(xi.maximum_consecutive_tx_days(tx, 290.0, ta, freq, before, **kw),
                                          oidx.longest_run_index(tx, ">", 290.0, 1, ot, freq, before))
        np.testing.assert_allclose(xi.growing_degree_days(tas, 277.15, ta, freq, **kw), ogen.cumulative_difference(tas, 277.15, ">", ot, freq),
                                   rtol=1e-6, atol=1e-4)
        np.testing.assert_allclose(xi.heating_degree_days(tas, 290.15, ta, freq, **kw), ogen.cumulative_difference(tas, 290.15, "<", ot, freq),
                                   rtol=1e-6, atol=1e-4)
        np.testing.assert_allclose(xi.daily_temperature_range(tn, tx, ta, freq, "mean", **kw),
                                   ogen.diurnal_temperature_range(tn, tx, "mean", ot, freq), rtol=1e-6, equal_nan=True)
        np.testing.assert_allclose(xi.daily_temperature_range_variability(tn, tx, ta, freq, **kw),
                                   ogen.interday_diurnal_temperature_range(tn, tx, ot, freq), rtol=1e-6, equal_nan=True)
        np.testing.assert_allclose(xi.extreme_temperature_range(tn, tx, ta, freq, **kw),
                                   ogen.extreme_temperature_range(tn, tx, ot, freq), rtol=1e-6, equal_nan=True)
    cond = ogen.compare(tas, ">=", 278.15)
    es, ee, el = orl.season_per_period(cond, 6, "07-01", ot, "YS")
    np.testing.assert_array_equal(xi.growing_season_length(tas, 278.15, ta, 6, "07-01", "YS", device=dev), el)
    with pytest.raises(ValueError):
        xi.tx_days_above(tx, 295.0, ta, "YS", "<", device=dev)
    # the MissingAny mask at index level: a period with a NaN day is NaN
    masked = xi.hot_spell_frequency(tx, 292.0, ta, 3, "MS", device=dev)
    seg, _ = ta.segments("MS")
    has_nan = np.stack([np.isnan(tx[a:b]).any(axis=0) for a, b in zip(seg[:-1], seg[1:])])
    incomplete = (np.diff(seg) != ta.expected_count("MS")).reshape((-1,) + (1,) * (tx.ndim - 1))  # e.g. a partial last month
    np.testing.assert_array_equal(np.isnan(masked), has_nan | incomplete)


def _precip(rng, T, shape, nan_frac=0.0):
    """kg m-2 s-1 daily precipitation: ~55 % dry days, gamma-distributed wet days."""
    wet = rng.random((T,) + shape) < 0.45
    x = np.where(wet, rng.gamma(0.8, 6.0, (T,) + shape) / 86400.0, 0.0).astype(np.float32)
    if nan_frac:
        x[rng.random(x.shape) < nan_frac] = np.nan
    return x


@pytest.mark.parametrize("calendar,T", [("standard", 1461), ("noleap", 1095)])
@pytest.mark.parametrize("op", [">", ">="])
def test_precip_percentile_indices(dev, rng, calendar, T, op):
    """days_over_precip_thresh / fraction_over_precip_thresh (indices/_multivariate.py:1174-1296): one
    fused pass against max(percentile_doy row, wet-day threshold); counts bit-exact, fractions <= 1e-6 relative."""
    pr = _precip(rng, T, (5, 6), nan_frac=0.003)
    pr[:, 0, 0] = 0.0        # never wet: 0 / 0 -> NaN fraction
    ta, ot = _axes("2000-01-01", T, calendar)
    thresh = 1.0 / 86400.0   # the reference default "1 mm/day" in kg m-2 s-1; not representable in float32
    p = percentile_doy(pr, ta, window=5, per=[75.0, 95.0], device=dev)
    exp, doys = ocal.percentile_doy(pr, ot, 5, [75.0, 95.0])
    for freq in ("YS", "MS"):
        for j, q in enumerate((75.0, 95.0)):
            got = xi.days_over_precip_thresh(pr, p.sel(q), ta, freq, op, thresh=thresh, device=dev, mask_missing=False)
            ref = oidx.days_over_precip_thresh(pr, exp[..., j], doys, ot, thresh, freq, op)
            np.testing.assert_array_equal(got, ref)
            got = xi.fraction_over_precip_thresh(pr, p.sel(q), ta, freq, op, thresh=thresh, device=dev, mask_missing=False)
            ref = oidx.fraction_over_precip_thresh(pr, exp[..., j], doys, ot, thresh, freq, op)
            np.testing.assert_allclose(got, ref, rtol=1e-6, equal_nan=True)
            assert np.isnan(got[:, 0, 0]).all() and np.nanmax(got) <= 1.0 and np.nanmin(got) >= 0.0
    # a per-cell percentile (pr.quantile(q, dim="time") in the reference's tests/test_indices.py:1595-1614)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        cell = np.nanquantile(np.where(pr > thresh, pr, np.nan).astype(np.float64), 0.9, axis=0)
    got = xi.days_over_precip_thresh(pr, cell, ta, "YS", op, thresh=thresh, device=dev, mask_missing=False)
    np.testing.assert_array_equal(got, oidx.days_over_precip_thresh(pr, cell, None, ot, thresh, "YS", op))
    got = xi.fraction_over_precip_thresh(pr, cell, ta, "MS", op, thresh=thresh, device=dev, mask_missing=False)
    np.testing.assert_allclose(got, oidx.fraction_over_precip_thresh(pr, cell, None, ot, thresh, "MS", op), rtol=1e-6, equal_nan=True)
    # MissingAny
    masked = xi.days_over_precip_thresh(pr, p.sel(75.0), ta, "MS", op, thresh=thresh, device=dev)
    seg, _ = ta.segments("MS")
    has_nan = np.stack([np.isnan(pr[a:b]).any(axis=0) for a, b in zip(seg[:-1], seg[1:])])
    np.testing.assert_array_equal(np.isnan(masked), has_nan)
    with pytest.raises(ValueError):
        xi.days_over_precip_thresh(pr, p.sel(75.0), ta, "YS", "<", thresh=thresh, device=dev)


def test_precip_percentile_indices_reference_known_answers(dev):
    """tests/test_indices.py:1579-1614 (TestDaysOverPrecipThresh) through the HIP path."""
    from xclim_amd.calendar import DoyPercentile

    a = np.zeros(365, dtype=np.float32)
    a[:8] = np.arange(8)
    ta = TimeAxis.daily("2000-01-01", 365)
    per = np.zeros(366)
    per[5:] = 5
    table = dev.to_device(per.reshape(1, 366, 1))
    p = DoyPercentile(table, np.arange(1, 367), [50.0], (), {})
    assert xi.days_over_precip_thresh(a[:, None], p, ta, thresh=2.0, device=dev, mask_missing=False)[0, 0] == 4
    f = xi.fraction_over_precip_thresh(a[:, None], p, ta, thresh=2.0, device=dev, mask_missing=False)
    np.testing.assert_array_almost_equal(f[0, 0], (3 + 4 + 6 + 7) / (3 + 4 + 5 + 6 + 7))
    assert xi.days_over_precip_thresh(a[:, None], np.array([5.0]), ta, thresh=2.0, device=dev, mask_missing=False)[0, 0] == 2
    t300 = TimeAxis.daily("2000-01-01", 300)
    out = xi.days_over_precip_thresh(np.ones((300, 2, 3), np.float32), np.zeros((2, 3)), t300, thresh=0.5, device=dev,
                                     mask_missing=False)
    np.testing.assert_array_equal(out, np.full((1, 2, 3), 300))
    with pytest.raises(KeyError):  # tests/test_bootstrapping.py:77-86: bootstrap needs a day-of-year percentile
        xi.days_over_precip_thresh(a[:, None], np.array([5.0]), ta, thresh=2.0, device=dev, bootstrap=True)


@pytest.mark.parametrize("stat", ["count", "frac"])
def test_precip_percentile_bootstrap(dev, rng, stat):
    """tests/test_bootstrapping.py:38-40: days_over / fraction_over_precip_thresh with bootstrap=True ("MS")."""
    from oracle import bootstrapping as oboot

    T = 1461
    pr = _precip(rng, T, (3, 4))
    ta, ot = _axes("2000-01-01", T)
    thresh = 1.0 / 86400.0
    base = (2000, 2001)
    b = ta.year <= 2001
    nb = int(b.sum())
    p = percentile_doy(pr[:nb], ta.subset(slice(0, nb)), window=5, per=98.0, device=dev)
    f = xi.days_over_precip_thresh if stat == "count" else xi.fraction_over_precip_thresh
    of = oidx.days_over_precip_thresh if stat == "count" else oidx.fraction_over_precip_thresh
    got = f(pr, p, ta, "MS", thresh=thresh, device=dev, bootstrap=True, mask_missing=False)
    exp = oboot.bootstrap_exceedance(pr, ot, base, "MS", ">", 5, 98.0,
                                     index_fn=lambda x, pp, dd, t: of(x, pp, dd, t, thresh, "MS", ">"))
    np.testing.assert_allclose(got, exp, rtol=1e-6, atol=0, equal_nan=True)
    plain = f(pr, p, ta, "MS", thresh=thresh, device=dev, mask_missing=False)
    out_base = ta.segments("MS")[0][:-1] >= nb
    np.testing.assert_allclose(got[out_base], plain[out_base], rtol=1e-6, equal_nan=True)


@pytest.mark.parametrize("before", [True, False])
@pytest.mark.parametrize("op", [">", ">="])
def test_heat_wave_indices(dev, rng, before, op):
    """heat_wave_frequency / max_length / total_length (indices/_multivariate.py:640-862): bivariate daily condition
    -> resample_and_rl(windowed_run_events | rle_statistics max | windowed_run_count); bit-exact."""
    T = 1095
    tn = _temp(rng, T, (5, 6), nan_frac=0.003) - 5.0
    slow = np.repeat(rng.normal(0, 4.0, (T // 6 + 1, 5, 6)), 6, axis=0)[:T].astype(np.float32)
    tn += slow
    tx = tn + 8.0 + rng.normal(0, 1.5, tn.shape).astype(np.float32)
    tx[rng.random(tx.shape) < 0.003] = np.nan
    ta, ot = _axes("2001-01-01", T)
    a, b = 22.0 + 273.15 - 8, 30.0 + 273.15 - 8   # not representable in float32
    total = 0
    for freq in ("YS", "MS", "QS-DEC"):
        for window in (1, 3, 5):
            for xf, of in ((xi.heat_wave_frequency, oidx.heat_wave_frequency), (xi.heat_wave_max_length, oidx.heat_wave_max_length),
                           (xi.heat_wave_total_length, oidx.heat_wave_total_length)):
                got = xf(tn, tx, ta, a, b, window, freq, op, before, device=dev, mask_missing=False)
                ref = of(tn, tx, ot, a, b, window, freq, op, before)
                np.testing.assert_array_equal(got, ref)
                total += got.sum()
    assert total > 0
    masked = xi.heat_wave_total_length(tn, tx, ta, a, b, 3, "MS", op, before, device=dev)
    seg, _ = ta.segments("MS")
    has_nan = np.stack([(np.isnan(tn[s:e]) | np.isnan(tx[s:e])).any(axis=0) for s, e in zip(seg[:-1], seg[1:])])
    np.testing.assert_array_equal(np.isnan(masked), has_nan)
    with pytest.raises(ValueError):
        xi.heat_wave_frequency(tn, tx, ta, a, b, op="<", device=dev)


def test_heat_wave_indices_reference_known_answers(dev):
    """tests/test_indices.py:1859-1960 through the HIP path."""
    K2C = 273.15
    tn = (np.asarray([20, 23, 23, 23, 23, 22, 23, 23, 23, 23]) + K2C).astype(np.float32)[:, None]
    tx = (np.asarray([29, 31, 31, 31, 29, 31, 31, 31, 31, 31]) + K2C).astype(np.float32)[:, None]
    ta = TimeAxis.daily("2000-01-01", 10)
    kw = dict(device=dev, mask_missing=False)
    for (a, b, w), f, m, tot in [((22, 30, 3), 2, 4, 7), ((10, 10, 3), 1, 10, 10), ((40, 40, 3), 0, 0, 0), ((22, 30, 4), 1, 4, 4),
                                 ((22, 30, 5), 0, 0, 0)]:
        assert xi.heat_wave_frequency(tn, tx, ta, a + K2C, b + K2C, w, **kw)[0, 0] == f
        assert xi.heat_wave_max_length(tn, tx, ta, a + K2C, b + K2C, w, **kw)[0, 0] == m
        assert xi.heat_wave_total_length(tn, tx, ta, a + K2C, b + K2C, w, **kw)[0, 0] == tot


@pytest.mark.parametrize("before", [True, False])
def test_hot_spell_max_magnitude(dev, rng, before):
    """indices/_threshold.py:2019-2066: (tasmax - thresh).clip(0) -> windowed_max_run_sum, cut at the period edges or
    resampled after; the reference's known answer (tests/test_indices.py:2132-2142) and seeded parity."""
    a = np.zeros(365)
    a[15:20] += 30
    a[40:42] += 50
    a[86:96] += 30
    da = (a + 273.15).astype(np.float32)[:, None]
    ta = TimeAxis.daily("2000-07-01", 365)
    out = xi.hot_spell_max_magnitude(da, 25 + 273.15, ta, 3, "ME", device=dev, mask_missing=False)
    np.testing.assert_allclose(out[:, 0], [25, 0, 30, 20, 0, 0, 0, 0, 0, 0, 0, 0], atol=1e-3)
    T = 1095
    x = _temp(rng, T, (5, 6), nan_frac=0.003)
    x += np.repeat(rng.normal(0, 3.0, (T // 7 + 1, 5, 6)), 7, axis=0)[:T].astype(np.float32)
    ta, ot = _axes("2001-01-01", T)
    for freq in ("YS", "MS"):
        for window in (1, 3):
            got = xi.hot_spell_max_magnitude(x, 295.15, ta, window, freq, before, device=dev, mask_missing=False)
            ref = oidx.hot_spell_max_magnitude(x, 295.15, ot, window, freq, before)
            np.testing.assert_allclose(got, ref, rtol=1e-6, atol=0)
    assert np.nanmax(got) > 0


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("before", [True, False])
def test_per_cell_thresholds(dev, rng, dtype, before):
    """A threshold with one value per grid cell (a DataArray threshold in the reference, e.g. a percentile over time):
    threshold_count, spell_length_statistics and the hot-spell indices compare in place against a one-row table; numpy
    broadcasting in the oracle.  float32 thresholds compare in float32, float64 ones widen the data."""
    T = 1095
    x = _temp(rng, T, (5, 6), nan_frac=0.003)
    x += np.repeat(rng.normal(0, 3.0, (T // 7 + 1, 5, 6)), 7, axis=0)[:T].astype(np.float32)
    ta, ot = _axes("2001-01-01", T)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        thr = np.nanquantile(x.astype(np.float64), 0.8, axis=0).astype(dtype)
    thr[0, 0] = x[5, 0, 0]   # exact ties with the data
    for freq in ("YS", "MS"):
        got = xgen.threshold_count(x, ">", thr, ta, freq, device=dev)
        np.testing.assert_array_equal(got, ogen.threshold_count(x, ">", thr, ot, freq))
        got = xgen.threshold_count(x, "<=", thr[None], ta, freq, device=dev)
        np.testing.assert_array_equal(got, ogen.threshold_count(x, "<=", thr[None], ot, freq))
        for red in ("max", "sum", "count"):
            got = xgen.spell_length_statistics(x, thr, 1, None, ">", red, ta, freq, resample_before_rl=before, device=dev)
            ref = ogen.spell_length_statistics(x, thr, 1, None, ">", red, ot, freq, resample_before_rl=before)
            np.testing.assert_array_equal(got, ref)
        got = xi.hot_spell_frequency(x, thr, ta, 3, freq, ">", before, device=dev, mask_missing=False)
        np.testing.assert_array_equal(got, oidx.run_index(x, ">", thr, "events", 3, ot, freq, before))
        got = xi.hot_spell_total_length(x, thr, ta, 3, freq, ">", before, device=dev, mask_missing=False)
        np.testing.assert_array_equal(got, oidx.run_index(x, ">", thr, "count", 3, ot, freq, before))
    assert got.sum() > 0


@pytest.mark.parametrize("calendar,T", [("standard", 1096), ("noleap", 730)])
def test_select_time_and_indexer(dev, rng, calendar, T):
    """calendar.select_time (cal:1259-1378) on the device: where-form (NaN outside) and drop-form, and the ``**indexer`` of
    select_resample_op (gen:109: ``da = select_time(da, **indexer)`` before the reduction)."""
    from xclim_amd.calendar import select_time

    x = _temp(rng, T, (4, 7), nan_frac=0.004)   # 28 cells: the scalar (non 16-byte) path; (T, 4, 8) below the vector path
    ta, ot = _axes("2001-01-01", T, calendar)
    for indexer in (dict(season="DJF"), dict(month=[6, 7, 8]), dict(doy_bounds=(340, 45)), dict(date_bounds=("03-15", "10-01")),
                    dict(doy_bounds=(100, 200), include_bounds=(False, True))):
        got = select_time(x, ta, device=dev, **indexer)
        np.testing.assert_array_equal(got, ocal.select_time(x, ot, **indexer))
        sub, tsub = select_time(x, ta, drop=True, device=dev, **indexer)
        esub, etsub = ocal.select_time(x, ot, drop=True, **indexer)
        np.testing.assert_array_equal(sub, esub)
        np.testing.assert_array_equal(tsub.doy, np.asarray(etsub.doy))
        for op in ("mean", "max", "sum", "count", "std"):
            got = xgen.select_resample_op(x, op, ta, "YS", device=dev, **indexer)
            ref = ogen.select_resample_op(ocal.select_time(x, ot, **indexer), op, ot, "YS")
            np.testing.assert_allclose(got, ref, rtol=1e-6, equal_nan=True)
    x8 = _temp(rng, T, (4, 8))
    np.testing.assert_array_equal(select_time(x8, ta, device=dev, month=1), ocal.select_time(x8, ot, month=1))
    assert select_time(x, ta, device=dev) is not None and np.array_equal(select_time(x, ta, device=dev), x, equal_nan=True)
    with pytest.raises(ValueError):
        select_time(x, ta, device=dev, month=1, season="DJF")


def test_indicator_level_time_selection(dev, rng):
    """Indicator-level ``**indexer`` (core/indicator.py + core/missing.py:118-135): every input is masked by select_time
    before the compute and MissingAny expects the selected days only."""
    T = 1096
    x = _temp(rng, T, (4, 6))
    x[rng.random(x.shape) < 0.002] = np.nan
    ta, ot = _axes("2001-01-01", T)

    def missing(data_m, expected):
        valid = np.stack([(~np.isnan(data_m[idx])).sum(axis=0) for _, idx in __import__("oracle.timeutil", fromlist=["groups"]).groups(ot, "YS")])
        return valid != np.asarray(expected).reshape(-1, 1, 1)

    for indexer, nsel in ((dict(season="JJA"), [92, 92, 92, 0]), (dict(month=[1, 2]), [59, 59, 59, 60]),
                          (dict(date_bounds=("11-15", "12-20")), [36, 36, 36, 0])):
        xm = ocal.select_time(x, ot, **indexer)
        # the last "year" holds 2004-01-01 only: its expected count is the selection inside the FULL year 2004
        exp_cnt = ta.expected_count("YS", **indexer)
        miss = missing(xm, exp_cnt)
        got = xi.tg_mean(x, ta, "YS", device=dev, **indexer)
        ref = ogen.select_resample_op(xm, "mean", ot, "YS").astype(np.float64)
        ref[miss] = np.nan
        np.testing.assert_allclose(got, ref, rtol=1e-6, equal_nan=True)
        got = xi.tx_days_above(x, 295.0, ta, "YS", device=dev, **indexer)
        ref = oidx.count_days(xm, ">", np.float32(295.0), ot, "YS").astype(np.float64)
        ref[miss] = np.nan
        np.testing.assert_array_equal(got, ref)
        got = xi.hot_spell_frequency(x, 293.0, ta, 2, "YS", device=dev, **indexer)
        ref = oidx.run_index(xm, ">", np.float32(293.0), "events", 2, ot, "YS").astype(np.float64)
        ref[miss] = np.nan
        np.testing.assert_array_equal(got, ref)
        assert list(exp_cnt[:3]) == nsel[:3]
    assert np.isfinite(xi.tg_mean(x, ta, "YS", device=dev, season="JJA")[:3]).any()
    # no selection: unchanged behaviour
    np.testing.assert_array_equal(xi.tg_mean(x, ta, "YS", device=dev), xi.tg_mean(x, ta, "YS", device=dev, season=None))


def test_missing_any_standalone(dev, rng):
    """xclim_amd.missing.missing_any == core/missing.py MissingAny (oracle.indices.missing_any), with and without a time
    selection; an incomplete last period is always missing."""
    from xclim_amd import missing as xmiss

    T = 800
    x = _temp(rng, T, (5, 4), nan_frac=0.002)
    ta, ot = _axes("2001-01-01", T)
    for freq in ("YS", "MS", "QS-DEC"):
        np.testing.assert_array_equal(xmiss.missing_any(x, freq, ta, device=dev), oidx.missing_any(x, ot, freq))
    got = xmiss.missing_any(x, "YS", ta, device=dev, season="JJA")
    xm = ocal.select_time(x, ot, season="JJA")
    seg, _ = ta.segments("YS")
    valid = np.stack([(~np.isnan(xm[a:b])).sum(axis=0) for a, b in zip(seg[:-1], seg[1:])])
    np.testing.assert_array_equal(got, valid != np.array([92, 92, 92]).reshape(-1, 1, 1))
    assert got[-1].all()   # 2003 stops in March: no JJA day at all


def test_dry_and_wet_spell_indices(dev, rng):
    """dry_spell_* / wet_spell_* (indices/_threshold.py:3314-3735) through the HIP path: the reference's known answers
    (tests/test_indices.py:4067-4171, incl. the date_bounds indexer that masks the SPELL MASK and relies on the 1-D
    ufunc semantics a single series gets) and seeded parity on windows that take the fused kernel."""
    from tests.test_oracle_reference_answers import _DRY_FREQ_OP, _DRY_SPELL_CASES

    kw = dict(device=dev, mask_missing=False)
    for pr, th1, th2, window, outs in _DRY_SPELL_CASES:
        x = np.asarray(pr, dtype=np.float32)[:, None]
        ta = TimeAxis.daily("1981-01-01", len(x))
        got = (xi.dry_spell_frequency(x, ta, th1, window, "YS", **kw)[0, 0],
               xi.dry_spell_total_length(x, ta, th2, window, "sum", "YS", **kw)[0, 0],
               xi.dry_spell_total_length(x, ta, th1, window, "max", "YS", **kw)[0, 0],
               xi.dry_spell_max_length(x, ta, th2, window, "sum", "YS", **kw)[0, 0],
               xi.dry_spell_max_length(x, ta, th1, window, "max", "YS", **kw)[0, 0])
        np.testing.assert_allclose(got, outs, rtol=1e-1)
    x = np.asarray([1] * 5 + [0] * 10 + [1] * 350, dtype=np.float32)[:, None]
    ta = TimeAxis.daily("1900-01-01", len(x), "noleap")
    for f in (xi.dry_spell_total_length, xi.dry_spell_max_length):
        out = f(x, ta, 3.1, 7, "sum", "MS", date_bounds=("01-10", "12-31"), **kw)
        np.testing.assert_allclose(out[:, 0], [9] + [0] * 11)
    x = np.asarray(_DRY_FREQ_OP, dtype=np.float32)[:, None]
    ta = TimeAxis.daily("2000-07-01", len(x))
    assert xi.dry_spell_frequency(x, ta, 1.0, 3, "MS", op="sum", **kw)[0, 0] == 2
    assert xi.dry_spell_frequency(x, ta, 1.0, 3, "MS", op="max", **kw)[0, 0] == 3
    T = 1095
    pr = (rng.gamma(0.7, 4.0, (T, 5, 6)) * (rng.random((T, 5, 6)) < 0.45)).astype(np.float32)
    pr[rng.random(pr.shape) < 0.002] = np.nan
    ta, ot = _axes("2001-01-01", T)
    for before in (True, False):
        for window, wop in ((3, "sum"), (5, "max"), (1, "sum"), (10, "sum")):
            for f, op, red in ((xi.dry_spell_frequency, "<", "count"), (xi.wet_spell_frequency, ">=", "count")):
                got = f(pr, ta, 1.0, window, "YS", before, wop, **kw)
                ref = ogen.spell_length_statistics(pr, np.float32(1.0), window, wop, op, red, ot, "YS", resample_before_rl=before)
                np.testing.assert_array_equal(got, ref)
            for f, op, red in ((xi.dry_spell_total_length, "<", "sum"), (xi.wet_spell_max_length, ">=", "max")):
                got = f(pr, ta, 1.0, window, wop, "MS", before, **kw)
                ref = ogen.spell_length_statistics(pr, np.float32(1.0), window, wop, op, red, ot, "MS", resample_before_rl=before)
                np.testing.assert_array_equal(got, ref)


def test_hot_and_cold_spell_reference_known_answers(dev):
    """tests/test_indices.py:119-147, 2040-2131 through the HIP path (fused compare + run statistics)."""
    from tests.test_oracle_reference_answers import _HS_FREQ, _HS_LEN, K2C, _spell_series

    kw = dict(device=dev, mask_missing=False)
    t10 = TimeAxis.daily("2000-07-01", 10)
    tx = (np.asarray(_HS_FREQ[0]) + K2C).astype(np.float32)[:, None]
    for th, w, op, exp in _HS_FREQ[1]:
        assert xi.hot_spell_frequency(tx, th + K2C, t10, w, "YS", op, **kw)[0, 0] == exp
    tx = (np.asarray(_HS_LEN[0]) + K2C).astype(np.float32)[:, None]
    for th, w, op, mx, tot in _HS_LEN[1]:
        assert xi.hot_spell_max_length(tx, th + K2C, t10, w, "YS", op, **kw)[0, 0] == mx
        assert xi.hot_spell_total_length(tx, th + K2C, t10, w, "YS", op, **kw)[0, 0] == tot
    hot, _, cold_f, order = _spell_series()
    t = TimeAxis.daily("2000-07-01", 365)
    np.testing.assert_array_equal(xi.hot_spell_total_length(hot[:, None], 25 + K2C, t, 5, "ME", **kw)[:, 0], [10, 0, 12, 8] + [0] * 8)
    cold = (2 * K2C - hot.astype(np.float64)).astype(np.float32)
    np.testing.assert_array_equal(xi.cold_spell_days(cold[:, None], -10 + K2C, t, 5, "ME", **kw)[:, 0], [10, 0, 12, 8] + [0] * 8)
    t71 = TimeAxis.daily("1971-01-01", 365)
    np.testing.assert_array_equal(xi.cold_spell_frequency(cold_f[:, None], -10 + K2C, t71, 5, "ME", **kw)[:, 0], [1, 0, 1, 1] + [0] * 8)
    assert xi.cold_spell_frequency(cold_f[:, None], -10 + K2C, t71, 5, "YS", **kw)[0, 0] == 3
    assert xi.hot_spell_frequency(order[:, None], 30 + K2C, t, 3, "MS", ">", True, **kw)[1, 0] == 1
    assert xi.hot_spell_frequency(order[:, None], 30 + K2C, t, 3, "MS", ">", False, **kw)[1, 0] == 0


def test_growing_season_length_reference_known_answers(dev):
    """tests/test_indices.py:1681-1707 through the HIP path (generic.season -> xh_season)."""
    from tests.test_oracle_reference_answers import _GSL_CASES, _gsl_series

    ta = TimeAxis.daily("2000-01-01", 365)
    for d1, d2, exp in _GSL_CASES:
        got = xi.growing_season_length(_gsl_series(d1, d2)[:, None], 278.15, ta, device=dev)
        assert got[0, 0] == exp, (d1, d2, got)
    ta2 = TimeAxis.daily("2000-01-01", 730)
    got = xi.growing_season_length(_gsl_series("2000-11-01", "2001-03-01", T=730)[:, None], 278.15, ta2, mid_date="01-01",
                                   freq="YS-JUL", device=dev)
    assert got[1, 0] == 121


def test_degree_days_reference_known_answers(dev):
    """tests/test_indices.py:232-247 (cooling_degree_days: 0 and 10), :1617-1622 (growing_degree_days: 1), :1835-1842
    (heating_degree_days: 6) through the HIP path (xh_thresholded_reduce), and against the oracle's cumulative_difference."""
    K2C = 273.15
    kw = dict(device=dev, mask_missing=False)
    t4 = TimeAxis.daily("2000-07-01", 4)
    a = (np.array([10, 15, -5, 18]) + K2C).astype(np.float32)[:, None]
    assert xi.cooling_degree_days(a, 18 + K2C, t4, **kw)[0, 0] == 0
    a = (np.array([20, 25, -15, 19]) + K2C).astype(np.float32)[:, None]
    np.testing.assert_allclose(xi.cooling_degree_days(a, 18 + K2C, t4, **kw)[0, 0], 10, rtol=1e-5)
    np.testing.assert_allclose(ogen.cumulative_difference(a, np.float32(18 + K2C), ">", OTime.standard("2000-07-01", 4), "YS")[0, 0], 10, rtol=1e-5)
    t = TimeAxis.daily("2000-07-01", 365)
    g = np.zeros(365)
    g[0] = 5
    np.testing.assert_allclose(xi.growing_degree_days((g + K2C).astype(np.float32)[:, None], 4 + K2C, t, **kw)[0, 0], 1, rtol=1e-4)
    h = np.zeros(365) + 17
    h[:7] += [-3, -2, -1, 0, 1, 2, 3]
    out = xi.heating_degree_days((h + K2C).astype(np.float32)[:, None], 17 + K2C, t, **kw)
    np.testing.assert_allclose(out[0, 0], 6, rtol=1e-4)
    assert (out[1:] == 0).all()


def test_generic_reference_known_answers(dev):
    """tests/test_generic.py:316-400, 769-797 through the HIP path: cumulative_difference ([0, 5, 10, 0, 0] / [20, 0, 0, 7,
    0] per daily period), first_day_threshold_reached for every operator, bivariate_spell_length_statistics (one spell,
    sum == max)."""
    K2C = 273.15
    tas = (np.array([-10, 15, 20, 3, 10]) + K2C).astype(np.float32)[:, None]
    t5 = TimeAxis.daily("2000-07-01", 5)
    for op, exp in ((">", [0, 5, 10, 0, 0]), (">=", [0, 5, 10, 0, 0]), ("<", [20, 0, 0, 7, 0])):
        # the reference calls it without freq (whole series reduced per time step group "D"): one value per day here
        got = [xgen.cumulative_difference(tas[i:i + 1], 10 + K2C, op, TimeAxis.daily("2000-07-01", 1), "YS", device=dev)[0, 0]
               for i in range(5)]
        np.testing.assert_allclose(got, exp, atol=1e-4)
    assert np.isclose(xgen.cumulative_difference(tas, 10 + K2C, ">", t5, "YS", device=dev)[0, 0], 15, atol=1e-4)
    with pytest.raises(NotImplementedError):
        xgen.cumulative_difference(tas, 10 + K2C, "!=", t5, "YS", device=dev)
    a = np.zeros(365, np.float32)
    a[:8] = (np.arange(8) / 1000).astype(np.float32)
    ta = TimeAxis.daily("2000-01-01", 365)
    for op, exp in ((">", 6), (">=", 5), ("==", 5), ("!=", 1)):
        got = xgen.first_day_threshold_reached(a[:, None], threshold=0.004, op=op, after_date="01-01", time=ta, window=1, freq="YS", device=dev)
        assert got[0, 0] == exp, (op, got)
    b = np.zeros(365, np.float32)
    b[:8] = np.flip(np.arange(8) / 1000).astype(np.float32)
    for op, exp in (("lt", 5), ("le", 4), ("eq", 4), ("ne", 1)):
        got = xgen.first_day_threshold_reached(b[:, None], threshold=0.004, op=op, after_date="01-01", time=ta, window=1, freq="YS", device=dev)
        assert got[0, 0] == exp, (op, got)
    with pytest.raises(ValueError):
        xgen.first_day_threshold_reached(b[:, None], threshold=0.004, op=">", after_date="01-01", time=ta, window=1, freq="YS",
                                         constrain=("<", "<="), device=dev)
    tn = np.full((365, 1), 270.0, np.float32)
    ta1 = TimeAxis.daily("2001-01-01", 365)
    outc, outs, outm = xgen.bivariate_spell_length_statistics(tn, 0 + K2C, tn.copy(), 1 + K2C, 5, "min", "<", ["count", "sum", "max"],
                                                              ta1, "YS", device=dev)
    np.testing.assert_array_equal(outs, outm)
    np.testing.assert_allclose(outc, 1)


def test_select_resample_op_reference_known_answers(dev):
    """tests/test_generic.py:16-83 through the HIP path: select_resample_op with month / season indexers (31; min of DJF
    0 and 366; 31 + 29 with YS-DEC), select_rolling_resample_op (trailing 14-day mean -> yearly max; rolling max + DJF
    indexer -> 14, 367, 732; centred 3-day integral per month), threshold_count [50, 0], domain_count [10, 0]."""
    q = np.arange(1000, dtype=np.float32)[:, None]
    ta = TimeAxis.daily("2000-01-01", 1000)
    np.testing.assert_array_equal(xgen.select_resample_op(q, "count", ta, "YS", device=dev, month=3)[:, 0], 31)
    o = xgen.select_resample_op(q, "min", ta, "YS", device=dev, season="DJF")
    assert o[0, 0] == 0 and o[1, 0] == 366
    assert xgen.select_resample_op(q, "count", ta, "YS-DEC", device=dev, season="DJF")[0, 0] == 31 + 29
    n = 366 + 365 + 365
    q3 = np.arange(1, n + 1, dtype=np.float32)[:, None]
    ta3 = TimeAxis.daily("2000-01-01", n)
    o = xgen.select_rolling_resample_op(q3, "max", 14, ta3, False, "mean", "YS", device=dev)
    np.testing.assert_allclose(o[:, 0], [np.mean(np.arange(353, 367)), np.mean(np.arange(353 + 365, 367 + 365)),
                                         np.mean(np.arange(353 + 730, 367 + 730))], rtol=1e-6)
    o = xgen.select_rolling_resample_op(q3, "min", 14, ta3, False, "max", "YS", device=dev, season="DJF")
    np.testing.assert_array_equal(o[:, 0], [14, 367, 367 + 365])
    o = xgen.select_rolling_resample_op(q3, "max", 3, ta3, True, "sum", "MS", device=dev)   # "integral" = sum x 86400 s
    np.testing.assert_array_equal(o[:2, 0], [30 + 31 + 32, 3 * 29 + 30 + 31 + 32])
    ts = np.arange(365, dtype=np.float32)[:, None]
    t = TimeAxis.daily("2000-07-01", 365)
    np.testing.assert_array_equal(xgen.threshold_count(ts, "<", 50, t, "YS", device=dev)[:, 0], [50, 0])
    np.testing.assert_array_equal(xgen.domain_count(ts, 10, 20, t, "YS", device=dev)[:, 0], [10, 0])


def test_counting_indices_reference_known_answers(dev):
    """tests/test_generic.py:401-511 through the HIP path: get_daily_events, count_level_crossings for the four operator
    pairs (and its forbidden pairs), count_occurrences with `constrain`, first / last occurrence (NaN without one)."""
    K2C = 273.15
    out = xgen.get_daily_events(np.array([-10, 15, 20, np.nan, 10], np.float32)[:, None], 10.0, ">=", device=dev)[:, 0]
    np.testing.assert_array_equal(out, [0, 1, 1, np.nan, 1])
    tn = (np.array([-1, -3, 0, 5, 9, 1, 3]) + K2C).astype(np.float32)[:, None]
    tx = (np.array([5, 7, 3, 6, 13, 5, 4]) + K2C).astype(np.float32)[:, None]
    t7 = TimeAxis.daily("2000-07-01", 7)
    for op_high, op_low, exp in ((">", "<", 1), (">", "<=", 2), (">=", "<", 3), (">=", "<=", 4)):
        got = xgen.count_level_crossings(tn, tx, 5 + K2C, t7, "YS", op_low=op_low, op_high=op_high, device=dev)
        assert got[0, 0] == exp, (op_high, op_low, got)
    for op_high, op_low in (("<=", "<="), (">=", ">="), ("<", ">"), ("==", "!=")):
        with pytest.raises(ValueError):
            xgen.count_level_crossings(tn, tx, 5 + K2C, t7, "YS", op_low=op_low, op_high=op_high, device=dev)
    tas = (np.arange(10) + K2C).astype(np.float32)[:, None]
    t10 = TimeAxis.daily("2000-07-01", 10)
    for op, constrain, exp, fail in (("<", ("!=", "<"), 4, False), (">", (">", "<="), 5, False), (">=", (">=", "=="), 6, False),
                                     ("==", ("==", "!="), 1, False), ("==", (">", ">="), 1, True), ("!=", ("!=", ">"), 9, False),
                                     ("!=", (">", "=="), 9, True), ("%", ("%", "$", "@"), 0, True)):
        if fail:
            with pytest.raises(ValueError):
                xgen.count_occurrences(tas, 4 + K2C, op, t10, "YS", constrain=constrain, device=dev)
        else:
            assert xgen.count_occurrences(tas, 4 + K2C, op, t10, "YS", constrain=constrain, device=dev)[0, 0] == exp
    tas = (np.array([15, 12, 11, 12, 14, 13, 18, 11, 13]) + K2C).astype(np.float32)[:, None]
    t9 = TimeAxis.daily("2000-01-01", 9)
    for f, cases in ((xgen.first_occurrence, (("<", None, np.nan), ("<=", None, 3), ("!=", ("!=",), 1), ("==", ("==", "!="), 3))),
                     (xgen.last_occurrence, (("<", None, np.nan), ("<=", None, 8), ("!=", ("!=",), 9), ("==", ("==", "!="), 8)))):
        for op, constrain, exp in cases:
            np.testing.assert_array_equal(f(tas, 11 + K2C, op, t9, "YS", constrain, device=dev)[0, 0], exp)
        with pytest.raises(ValueError):
            f(tas, 11 + K2C, "==", t9, "YS", (">=", ">", "<"), device=dev)


def test_day_count_indices_reference_known_answers(dev):
    """tests/test_indices.py:2145-2200 (tn / tg days above / below, operators, forbidden operators) and :4211-4224
    (wetdays [5, 0, 0, 3, ...] with >= and [4, 0, 0, 2, ...] with >) through the HIP path."""
    K2C = 273.15
    kw = dict(device=dev, mask_missing=False)
    t = TimeAxis.daily("2000-07-01", 365)
    a = np.zeros(365)
    a[:6] += [27, 28, 29, 30, 31, 32]
    mn = (a + K2C).astype(np.float32)[:, None]
    out = xi.tn_days_above(mn, 30 + K2C, t, **kw)
    assert out[0, 0] == 2 and (out[1:] == 0).all()
    assert xi.tn_days_above(mn, 30 + K2C, t, op=">=", **kw)[0, 0] == 3
    with pytest.raises(ValueError):
        xi.tn_days_above(mn, 30 + K2C, t, op="<=", **kw)
    b = np.zeros(365)
    b[:6] -= [27, 28, 29, 30, 31, 32]
    mb = (b + K2C).astype(np.float32)[:, None]
    assert xi.tn_days_below(mb, -10 + K2C, t, **kw)[0, 0] == 6
    assert xi.tn_days_below(mb, -30 + K2C, t, **kw)[0, 0] == 2
    assert xi.tn_days_below(mb, -31 + K2C, t, op="<=", **kw)[0, 0] == 2
    with pytest.raises(ValueError):
        xi.tn_days_below(mb, 30 + K2C, t, op=">=", **kw)
    p = np.zeros(365)
    p[:7] += [4, 5.5, 6, 6, 2, 7, 5]
    p[100:106] += [1, 6, 7, 5, 2, 1]
    pr = p.astype(np.float32)[:, None]
    np.testing.assert_array_equal(xi.wetdays(pr, 5.0, t, "ME", **kw)[:, 0], [5, 0, 0, 3] + [0] * 8)
    np.testing.assert_array_equal(xi.wetdays(pr, 5.0, t, "ME", ">", **kw)[:, 0], [4, 0, 0, 2] + [0] * 8)


def test_consecutive_day_indices_reference_known_answers(dev):
    """tests/test_indices.py:186-215 (maximum_consecutive_frost_days: 1 / 0 / all year) and :2384-2391
    (maximum_consecutive_tx_days: 10 in the first month, 0 after) through the HIP path."""
    K2C = 273.15
    kw = dict(device=dev, mask_missing=False)
    t5 = TimeAxis.daily("2000-07-01", 5)
    a = (np.array([3, 4, 5, -1, 3]) + K2C).astype(np.float32)[:, None]
    assert xi.maximum_consecutive_frost_days(a, 0 + K2C, t5, "YS-JUL", **kw)[0, 0] == 1
    a = (np.array([3, 4, 5, 1, 3]) + K2C).astype(np.float32)[:, None]
    assert xi.maximum_consecutive_frost_days(a, 0 + K2C, t5, "YS-JUL", **kw)[0, 0] == 0
    a = (np.zeros(365) - 10 + K2C).astype(np.float32)[:, None]
    out = xi.maximum_consecutive_frost_days(a, 0 + K2C, TimeAxis.daily("2000-07-01", 365), "YS-JUL", **kw)
    assert out[0, 0] == 365
    b = np.zeros(365) + 273.15
    b[5:15] += 30
    out = xi.maximum_consecutive_tx_days(b.astype(np.float32)[:, None], 25 + K2C, TimeAxis.daily("2010-01-01", 365), "ME", **kw)
    assert out[0, 0] == 10 and (out[1:] == 0).all()


@pytest.mark.parametrize("kind", ["+", "*"])
@pytest.mark.parametrize("interp", ["nearest", "linear", "cubic"])
@pytest.mark.parametrize("T,cells", [(365, (7, 9)), (800, (33,)), (3000, (5,)), (10950, (3,)), (50, (4, 4)), (1, (3,)), (20000, (2,)), (32768, (2,))])
def test_qdm_adjust_matches_oracle(dev, rng, kind, interp, T, cells):
    """QuantileDeltaMapping.adjust == rank(pct) + interp_on_quantiles + apply_correction of the oracle (scipy rankdata /
    interp1d), every kernel variant (T 1 .. 10950), NaN samples, tied samples, both extrapolations.  Parity unpinned."""
    from xclim_amd import sdba as xsdba

    shape = (T,) + cells
    ref = rng.normal(10, 3, shape).astype(np.float32)
    hist = rng.normal(11, 4, shape).astype(np.float32)
    sim = rng.normal(13, 4, shape).astype(np.float32)
    if kind == "*":
        ref, hist, sim = np.abs(ref) + 1, np.abs(hist) + 1, np.abs(sim) + 1
    sim[rng.random(shape) < 0.03] = np.nan
    if T > 10:
        sim[T // 3] = sim[T // 2]          # ties (a whole row repeated)
        sim[:, ..., 0] = np.round(sim[:, ..., 0])   # a heavily tied cell
    qdm = xsdba.QuantileDeltaMapping.train(ref, hist, nquantiles=20, kind=kind, device=dev)
    for extrap in ("constant", "nan"):
        got = qdm.adjust(sim, interp=interp, extrapolation=extrap)
        exp = osdba.qdm_adjust(sim, qdm.af, qdm.quantiles, kind, interp, extrap)
        # (cubic, round 5: the spline is evaluated at the float32 percentage rank over float32 node abscissae — the factor
        # moves by its slope times ~1e-7; the same 2e-6 as the EQM cubic path)
        np.testing.assert_allclose(got, exp, rtol=2e-6 if interp == "cubic" else 1e-6, atol=0, equal_nan=True, err_msg=f"{extrap}")


@pytest.mark.parametrize("kind", ["+", "*"])
@pytest.mark.parametrize("group", ["time.month", "time.season"])
def test_dqm_grouped_matches_oracle(dev, rng, kind, group):
    """DetrendedQuantileMapping with a sub-grouping (round 3: group="time" only): per group the normalised quantiles and
    the scaling; adjust = the group's scaling, a trend fitted over the group's OWN steps on their time coordinate, the
    group's nodes, the trend put back.  A warming sim (trend 3 K over the series) keeps its trend; parity unpinned."""
    from xclim_amd import sdba as xsdba

    T = 365 * 6
    ta, ot = _axes("2001-01-01", T, "noleap")
    shape = (T, 3, 4)
    t = np.arange(T)[:, None, None]
    seas = 8 * np.sin(2 * np.pi * (t - 100) / 365)
    base = 0.0 if kind == "+" else 25.0
    ref = (base + 10 + seas + rng.normal(0, 3, shape)).astype(np.float32)
    hist = (base + 11.5 + 1.2 * seas + rng.normal(0, 4, shape)).astype(np.float32)
    sim = (base + 12 + 1.2 * seas + 3.0 * t / T + rng.normal(0, 4, shape)).astype(np.float32)
    sim[rng.random(shape) < 0.01] = np.nan
    hist[:50, 0, 0] = np.nan
    prop = group.split(".")[1]
    dqm = xsdba.DetrendedQuantileMapping.train(ref, hist, nquantiles=15, kind=kind, group=group, time=ta, device=dev)
    labels, eaf, ehq, escal = osdba.dqm_train_grouped(ref, hist, ot, prop, 15, kind)
    np.testing.assert_array_equal(dqm.group_labels, labels)
    np.testing.assert_allclose(dqm.scaling, escal, rtol=1e-6)
    np.testing.assert_allclose(dqm.hist_q, ehq, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(dqm.af, eaf, rtol=1e-5, atol=1e-5)
    for deg in (0, 1):
        for mode in ("griddata", "group"):
            got = dqm.adjust(sim, detrend=deg, time=ta, grouped_nearest=mode)
            exp = osdba.dqm_adjust_grouped(sim, ot, prop, labels, dqm.af, dqm.hist_q, dqm.scaling, kind, "constant", deg,
                                           mode=mode if prop == "month" else "group")
            np.testing.assert_allclose(got, exp, rtol=2e-5, atol=2e-5, equal_nan=True)
    first, last = np.nanmean(got[:365]), np.nanmean(got[-365:])
    assert 1.5 < last - first < 4.0                                     # the simulated trend survives the adjustment
    if prop == "month":   # interp="linear": the detrended steps interpolated over the (quantile, group) plane (round 5)
        got = dqm.adjust(sim, interp="linear", detrend=1, time=ta)
        exp = osdba.dqm_adjust_grouped(sim, ot, prop, labels, dqm.af, dqm.hist_q, dqm.scaling, kind, "constant", 1, mode="griddata",
                                       interp="linear")
        np.testing.assert_allclose(got, exp, rtol=2e-5, atol=2e-5, equal_nan=True)
    else:
        with pytest.raises(NotImplementedError):
            dqm.adjust(sim, interp="linear", time=ta)


@pytest.mark.parametrize("kind", ["+", "*"])
@pytest.mark.parametrize("group,window", [("time.dayofyear", 31), ("time.month", 5)])
def test_dqm_windowed_sub_grouping_matches_oracle(dev, rng, kind, group, window):
    """DetrendedQuantileMapping with a WINDOWED sub-grouping (round 5; the documented standard Grouper("time.dayofyear",
    window=31)): the training sample of a group is the windowed one (means over time and window), and PolyDetrend fits the
    group's trend on the centred window mean of the scaled series (xh_window_nanmean).  nearest (own group + plane) and
    linear over the (quantile, group) plane; parity unpinned."""
    from xclim_amd import kernels as K
    from xclim_amd import sdba as xsdba

    # the window mean itself: NaN samples, windows cut by the ends of the series, a window without any valid sample
    x = rng.normal(5, 3, (300, 70)).astype(np.float32)
    x[rng.random(x.shape) < 0.1] = np.nan
    x[100:140, 3] = np.nan
    x[:, 4] = np.nan
    for w in (1, 3, 31):
        got = K.window_nanmean(dev, dev.to_device(x), w).get()
        np.testing.assert_allclose(got, osdba.window_nanmean(x, w), rtol=3e-7, atol=1e-7, equal_nan=True)

    T = 365 * 4   # (the CPU oracle runs scipy.griddata once per group: 365 of them for the day of the year)
    ta, ot = _axes("2001-01-01", T, "noleap")
    shape = (T, 1, 3)
    t = np.arange(T)[:, None, None]
    seas = 8 * np.sin(2 * np.pi * (t - 100) / 365)
    base = 0.0 if kind == "+" else 25.0
    ref = (base + 10 + seas + rng.normal(0, 3, shape)).astype(np.float32)
    hist = (base + 11.5 + 1.2 * seas + rng.normal(0, 4, shape)).astype(np.float32)
    sim = (base + 12 + 1.2 * seas + 3.0 * t / T + rng.normal(0, 4, shape)).astype(np.float32)
    sim[rng.random(shape) < 0.01] = np.nan
    hist[:50, 0, 0] = np.nan
    prop = group.split(".")[1]
    dqm = xsdba.DetrendedQuantileMapping.train(ref, hist, nquantiles=15, kind=kind, group=group, window=window, time=ta, device=dev)
    labels, eaf, ehq, escal = osdba.dqm_train_grouped(ref, hist, ot, prop, 15, kind, window=window)
    np.testing.assert_array_equal(dqm.group_labels, labels)
    np.testing.assert_allclose(dqm.scaling, escal, rtol=1e-6)
    np.testing.assert_allclose(dqm.hist_q, ehq, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(dqm.af, eaf, rtol=1e-5, atol=1e-5)
    for deg, mode in ((0, "group"), (1, "group"), (1, "griddata")):
        if True:
            got = dqm.adjust(sim, detrend=deg, time=ta, grouped_nearest=mode)
            exp = osdba.dqm_adjust_grouped(sim, ot, prop, labels, dqm.af, dqm.hist_q, dqm.scaling, kind, "constant", deg, mode=mode,
                                           window=window)
            bad = ~np.isclose(got, exp, rtol=2e-5, atol=2e-5, equal_nan=True)
            # windowed groups SHARE samples, so neighbouring groups often hold the same extreme node: two nodes of the plane at
            # exactly the same distance of a query, of which scipy's cKDTree returns either (verified one by one in
            # test_grouper_add_dims_pools_the_members and tools/fuzz_plane.py; here: a handful at most, on some seeds)
            assert bad.sum() <= (3 if mode == "griddata" else 0), f"{deg} {mode}: {int(bad.sum())} mismatches"
    first, last = np.nanmean(got[:365]), np.nanmean(got[-365:])
    assert 1.5 < last - first < 4.0                                     # the simulated trend survives the adjustment
    got = dqm.adjust(sim, interp="linear", detrend=1, time=ta)
    exp = osdba.dqm_adjust_grouped(sim, ot, prop, labels, dqm.af, dqm.hist_q, dqm.scaling, kind, "constant", 1, mode="griddata",
                                   interp="linear", window=window)
    np.testing.assert_allclose(got, exp, rtol=2e-5, atol=2e-5, equal_nan=True)


@pytest.mark.parametrize("kind", ["+", "*"])
@pytest.mark.parametrize("group,window", [("time", 1), ("time.month", 1), ("time.dayofyear", 15)])
def test_grouper_add_dims_pools_the_members(dev, rng, kind, group, window):
    """Grouper(add_dims=...) (round 5; /root/reference/docs/sdba.rst:64-66: "the factors for each day of the year but across
    all realizations of an ensemble"): the members' samples are pooled with the time steps of a group when the quantiles
    are taken, the factors have no member axis, and adjust maps every member of sim with them — EQM and QDM; the oracle
    pools by reshaping (n, R, cells) -> (n R, cells).  Parity unpinned (xsdba)."""
    from xclim_amd import sdba as xsdba

    T, R = 365 * 3, 3
    ta, ot = _axes("2001-01-01", T, "noleap")
    shape = (T, R, 2, 3)
    t = np.arange(T)[:, None, None, None]
    seas = 8 * np.sin(2 * np.pi * (t - 100) / 365)
    member = np.arange(R)[None, :, None, None] * 0.7
    base = 0.0 if kind == "+" else 25.0
    ref = (base + 10 + seas + member + rng.normal(0, 3, shape)).astype(np.float32)
    hist = (base + 11.5 + 1.2 * seas + 2 * member + rng.normal(0, 4, shape)).astype(np.float32)
    sim = (base + 12 + 1.2 * seas + 2 * member + rng.normal(0, 4, shape)).astype(np.float32)
    sim[rng.random(shape) < 0.01] = np.nan
    hist[:40, 1, 0, 0] = np.nan
    prop = group.split(".")[1] if "." in group else "group"
    grp = xsdba.Grouper(group, window, add_dims=1)
    assert "add_dims=(1,)" in repr(grp)
    q = osdba.equally_spaced_nodes(12)

    def pooled(x, lab):
        smp = x if prop == "group" else osdba.grouped_sample(x, ot, prop, window, lab)
        return smp.reshape((-1,) + x.shape[2:])

    labels = np.array([0]) if prop == "group" else np.unique(osdba.group_values(ot, prop))
    tabs = [osdba.eqm_train(pooled(ref, lab), pooled(hist, lab), q, kind) for lab in labels]
    eaf, ehq = np.stack([a for a, _ in tabs]), np.stack([h for _, h in tabs])
    for cls in (xsdba.EmpiricalQuantileMapping, xsdba.QuantileDeltaMapping):
        mdl = cls.train(ref, hist, nquantiles=12, kind=kind, group=grp, time=ta, device=dev)
        assert mdl.cell_shape == (2, 3)
        lead = () if prop == "group" else (len(labels),)
        np.testing.assert_allclose(mdl.hist_q, ehq.reshape(lead + (12, 2, 3)), rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(mdl.af, eaf.reshape(lead + (12, 2, 3)), rtol=1e-5, atol=1e-5)
        got = mdl.adjust(sim, time=ta)
        assert got.shape == shape
        for r in range(R):
            if cls is xsdba.EmpiricalQuantileMapping:
                exp = (osdba.eqm_adjust(sim[:, r], mdl.af, mdl.hist_q, kind, "nearest", "constant") if prop == "group" else
                       osdba.eqm_adjust_grouped(sim[:, r], ot, prop, labels, mdl.af, mdl.hist_q, kind, "nearest", "constant", mode="griddata"))
            else:
                exp = (osdba.qdm_adjust(sim[:, r], mdl.af, mdl.quantiles, kind, "nearest", "constant") if prop == "group" else
                       osdba.qdm_adjust_grouped(sim[:, r], ot, prop, labels, mdl.af, mdl.quantiles, kind, "nearest", "constant", mode="group"))
            bad = ~np.isclose(got[:, r], exp, rtol=1e-6, atol=1e-6, equal_nan=True)
            if bad.any() and cls is xsdba.EmpiricalQuantileMapping and prop == "dayofyear":
                bad = _accept_plane_nearest_ties(bad, got[:, r], sim[:, r], ot.doy, mdl.hist_q, mdl.af, kind, limit=4)
            assert not bad.any(), f"{cls.__name__} member {r}: {int(bad.sum())} mismatches"
        # a sim WITHOUT the member axis takes the same factors (the trained shape)
        np.testing.assert_array_equal(mdl.adjust(sim[:, 1], time=ta), got[:, 1])
    with pytest.raises(NotImplementedError):
        xsdba.Grouper("time.month", add_dims=2)            # not the axis right behind time
    # DetrendedQuantileMapping with the pooled members: group="time" (the trend is fitted on the member mean), refused for sub-groupings
    if prop == "group":
        dqm = xsdba.DetrendedQuantileMapping.train(ref, hist, nquantiles=12, kind=kind, group=grp, device=dev)
        eaf, ehq, escal = osdba.dqm_train(ref.reshape((-1,) + ref.shape[2:]), hist.reshape((-1,) + hist.shape[2:]), q, kind)
        np.testing.assert_allclose(dqm.scaling, escal, rtol=1e-6)
        np.testing.assert_allclose(dqm.hist_q, ehq, rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(dqm.af, eaf, rtol=1e-5, atol=1e-5)
        for deg in (0, 1):
            got = dqm.adjust(sim, detrend=deg)
            exp = osdba.dqm_adjust_members(sim, dqm.af, dqm.hist_q, dqm.scaling, kind, "nearest", "constant", deg, pooled=True)
            np.testing.assert_allclose(got, exp, rtol=2e-5, atol=2e-5, equal_nan=True, err_msg=f"detrend {deg}")
        plain = xsdba.DetrendedQuantileMapping.train(ref[:, 0], hist[:, 0], nquantiles=12, kind=kind, device=dev)
        got = plain.adjust(sim, interp="linear")      # an ordinary extra axis: every series keeps its own trend
        exp = osdba.dqm_adjust_members(sim, plain.af, plain.hist_q, plain.scaling, kind, "linear", "constant", 1, pooled=False)
        np.testing.assert_allclose(got, exp, rtol=2e-5, atol=2e-5, equal_nan=True)
    else:
        with pytest.raises(NotImplementedError):
            xsdba.DetrendedQuantileMapping.train(ref, hist, group=grp, time=ta, device=dev)


@pytest.mark.parametrize("T", [1, 2, 700, 40000])
def test_quantile_cells_per_cell_probabilities(dev, rng, T):
    """xh_quantile_cells = xsdba.nbutils.vecquantiles: one quantile per cell at its own probability; NaN probabilities
    (also with a single valid sample: found by tools/fuzz_r04.py), 0 and 1, all-NaN cells, both layouts."""
    from xclim_amd import kernels as K

    C = 37
    x = rng.normal(5, 3, (T, C)).astype(np.float32)
    x[rng.random((T, C)) < 0.05] = np.nan
    x[:, 2] = np.nan
    qc = rng.random(C)
    qc[[0, 5]], qc[7], qc[8] = np.nan, 0.0, 1.0
    with np.errstate(all="ignore"):
        exp = np.array([np.nanquantile(x[:, c].astype(np.float64), qc[c]) if qc[c] == qc[c] and not np.isnan(x[:, c]).all() else np.nan
                        for c in range(C)])
    got = K.quantile_cells(dev, dev.to_device(x), qc).get()
    np.testing.assert_allclose(got, exp, rtol=1e-6, equal_nan=True)
    got_t = K.quantile_cells(dev, dev.to_device(np.ascontiguousarray(x.T)), qc, time_axis=1).get()
    np.testing.assert_array_equal(got_t, got)


def _pr_field(rng, T, C, p_dry, scale):
    x = rng.gamma(0.7, scale, (T, C)).astype(np.float32)
    x[rng.random((T, C)) < p_dry] = 0.0
    return x


@pytest.mark.parametrize("group,window", [("time", 1), ("time.month", 1), ("time.dayofyear", 31)])
def test_adapt_freq_matches_oracle(dev, rng, group, window):
    """xsdba.processing.adapt_freq (round 3: not built): frequencies P0 / dP0, pth = vecquantiles(ref, P0_sim), the
    values whose percentage rank lies in [P0_ref, P0_sim] replaced by U[thresh, pth) — against the oracle's restatement
    with the SAME counter-based uniforms (upstream draws from numpy's global generator).  Cells: sim too dry (the case
    the method is for), sim wetter than ref (dP0 < 0: untouched), equal frequencies, no dry day at all (P0_sim = 0:
    dP0 NaN), NaN samples, all-NaN sim / ref, drizzle below the threshold (distinct ranks inside the band), -0.0."""
    from xclim_amd import sdba as xsdba

    T = 365 * 4
    ta, ot = _axes("2001-01-01", T, "noleap")
    C = 12
    ref, sim = _pr_field(rng, T, C, 0.4, 5.0), _pr_field(rng, T, C, 0.65, 4.0)
    sim[:, 1] = _pr_field(rng, T, 1, 0.2, 4.0)[:, 0]                 # wetter than ref
    sim[:, 2] = np.where(ref[:, 2] == 0, 0.0, sim[:, 2] + 0.5)       # the same dry days
    sim[:, 3] = np.abs(sim[:, 3]) + 1.5                               # never below the threshold
    sim[rng.random((T, C)) < 0.01] = np.nan
    ref[rng.random((T, C)) < 0.01] = np.nan
    sim[:, 4] = np.nan
    ref[:, 5] = np.nan
    dz = sim[:, 6] == 0
    sim[dz, 6] = rng.random(int(dz.sum())).astype(np.float32) * 0.9  # drizzle: distinct values below the threshold
    z = np.flatnonzero(sim[:, 7] == 0)
    sim[z[::2], 7] = -0.0
    thresh, seed = 1.0, 77
    prop = "group" if group == "time" else group.split(".")[1]
    got, pth, dp0 = xsdba.adapt_freq(ref, sim, thresh, group=group, window=window, time=ta, seed=seed, device=dev)
    exp, epth, edp0 = osdba.adapt_freq(ref, sim, thresh, seed=seed, time=ot, prop=prop, window=window)
    np.testing.assert_allclose(dp0, edp0, rtol=1e-12, equal_nan=True)
    np.testing.assert_allclose(pth, epth, rtol=1e-6, equal_nan=True)
    np.testing.assert_allclose(got, exp, rtol=1e-6, atol=0, equal_nan=True)
    changed = ~((got == sim) | (np.isnan(got) & np.isnan(sim)))
    assert changed[:, 0].sum() > 50 and not changed[:, 3].any() and not changed[:, 4].any()
    # ("wetter than ref" holds for the whole series; a single month or day-of-year group of it may be drier on some draws)
    assert group != "time" or not changed[:, 1].any()
    if group == "time":
        # the defining property: the frequency of values <= thresh in sim_ad matches ref's (up to the ties of exact zeros)
        p0 = lambda x: np.nanmean(np.where(np.isnan(x), np.nan, x <= thresh), axis=0)  # noqa: E731
        assert abs(p0(got)[6] - p0(ref)[6]) < 0.01 and p0(sim)[6] - p0(ref)[6] > 0.15
        assert np.all(got[changed[:, 6], 6] >= thresh) and np.all(got[changed[:, 6], 6] < pth[6])


def test_eqm_train_with_adapt_freq_thresh(dev, rng):
    """EmpiricalQuantileMapping.train(adapt_freq_thresh=...) (xsdba: eqm_train -> _adapt_freq_hist): hist is frequency-
    adapted against ref before the quantiles are taken, multiplicative factors stay finite at the dry nodes."""
    from xclim_amd import sdba as xsdba

    T, C = 3000, 9
    ref, hist = _pr_field(rng, T, C, 0.4, 5.0), _pr_field(rng, T, C, 0.7, 4.0)
    dz = hist == 0
    hist[dz] = rng.random(int(dz.sum())).astype(np.float32) * 0.4     # drizzle instead of tied zeros (xsdba: jitter_under_thresh first)
    eqm = xsdba.EmpiricalQuantileMapping.train(ref, hist, nquantiles=15, kind="*", adapt_freq_thresh=0.5, adapt_freq_seed=3, device=dev)
    hist_ad, _, _ = osdba.adapt_freq(ref, hist, 0.5, seed=3)
    eaf, ehq = osdba.eqm_train(ref, hist_ad, 15, "*")
    np.testing.assert_allclose(eqm.hist_q, ehq, rtol=1e-6, equal_nan=True)
    np.testing.assert_allclose(eqm.af, eaf, rtol=1e-5, equal_nan=True)
    plain = xsdba.EmpiricalQuantileMapping.train(ref, hist, nquantiles=15, kind="*", device=dev)
    # node 9 (q = 0.633) lies between the frequencies of values <= 0.5 in ref (~0.53) and hist (~0.78): drizzle before, wet after
    assert (plain.hist_q[9] < 0.5).all() and (eqm.hist_q[9] >= 0.5).all()


@pytest.mark.parametrize("kind,interp", [("+", "nearest"), ("*", "linear")])
def test_qdm_adjust_beyond_32768_steps(dev, rng, kind, interp):
    """1950-2100 daily (55 152 steps): round 3 refused series longer than 32768 steps.  qdm3.hip ranks them through a
    global sort of (key, time index) pairs; NaN samples, tied rows, a heavily tied cell, dry days, signed zeros, an
    all-NaN and a constant cell, NaN factors; both layouts.  Against the oracle (scipy rankdata / interp1d)."""
    from xclim_amd import kernels as K

    T, C, nq = 55152, 37, 20
    sim = rng.normal(13, 4, (T, C)).astype(np.float32)
    if kind == "*":
        sim = np.abs(sim) + 1
    sim[rng.random((T, C)) < 0.01] = np.nan
    sim[T // 3] = sim[T // 2]
    sim[:, 0] = np.round(sim[:, 0])
    sim[:, 1] = np.where(rng.random(T) < 0.6, 0.0, rng.gamma(0.7, 4.0, T)).astype(np.float32)
    z = np.flatnonzero(sim[:, 1] == 0.0)
    sim[z[::2], 1] = -0.0
    sim[:, 2] = np.nan
    sim[:, 3] = 7.5
    sim[1:, 4] = np.nan
    q = (np.arange(nq) + 0.5) / nq
    af = rng.normal(1.0, 0.3, (nq, C)).astype(np.float32)
    af[rng.random((nq, C)) < 0.02] = np.nan
    af[:, 6] = np.nan
    exp = osdba.qdm_adjust(sim, af, q, kind, interp, "constant")
    got = K.qdm_adjust(dev, dev.to_device(sim), dev.to_device(af), q, kind, interp, "constant").get()
    np.testing.assert_allclose(got, exp, rtol=1e-6, atol=0, equal_nan=True)
    tm = K.qdm_adjust(dev, dev.to_device(np.ascontiguousarray(sim.T)), dev.to_device(af), q, kind, interp, "constant", time_axis=1).get()
    np.testing.assert_array_equal(tm.T, got)


@pytest.mark.parametrize("T", [360, 365, 366])
@pytest.mark.parametrize("kind,nq", [("+", 20), ("*", 15), ("*", 36), ("+", 48)])
def test_qdm_nearest_one_year_cut_value_kernel(dev, rng, monkeypatch, T, kind, nq):
    """qdm2.hip (one-year series, interp="nearest": sort in registers, class boundaries as order statistics, samples
    classified against cut values) == the oracle AND bit-identical to the exact-rank kernel it replaces, on: continuous
    columns, NaN samples, dry-day series (copies of the minimum: the exact path inside the kernel), signed zeros, columns
    with ties (the list handed to the exact-rank kernel), constant and all-NaN columns, NaN factors (dropped nodes), both
    extrapolations, a ragged last tile."""
    from xclim_amd import kernels as K

    C = 203
    sim = rng.normal(13, 4, (T, C)).astype(np.float32)
    if kind == "*":
        sim = np.abs(sim) + 1
    sim[rng.random((T, C)) < 0.02] = np.nan
    dry = rng.random((T, 40)) < 0.6                                  # columns 0 .. 39: precipitation-like, 60 % dry
    sim[:, :40] = np.where(dry, 0.0, rng.gamma(0.7, 4.0, (T, 40))).astype(np.float32)
    z = np.flatnonzero(sim[:, 3] == 0.0)
    sim[z[::2], 3] = -0.0                                            # signed zeros tie
    sim[5:60, 4] = np.nan                                            # dry days and NaN samples together
    sim[:, 50] = np.round(sim[:, 50])                                # heavily tied column -> exact-rank kernel
    sim[10, 51] = sim[200, 51]                                       # one tie
    sim[:, 52] = 7.5                                                 # constant: 0 / 0 ranks -> NaN
    sim[:, 53] = np.nan
    sim[1:, 54] = np.nan                                             # a single valid sample
    sim[:, 55] = np.sort(sim[:, 55])                                 # sorted input (NaN last)
    sim[100, 56] = np.inf
    sim[101, 56] = -np.inf
    sim[[7, 300], 57] = np.nanmax(sim[:, 57]) + 1                    # two copies of the maximum: mx changes for every sample
    sim[[9, 301], 58] = np.nanmin(sim[:, 58]) - 1                    # two copies of the minimum of a continuous column
    q = (np.arange(nq) + 0.5) / nq
    af = rng.normal(1.0, 0.3, (nq, C)).astype(np.float32)
    af[rng.random((nq, C)) < 0.02] = np.nan                          # dropped nodes
    af[:, 60] = np.nan                                               # no node at all
    af[1:, 61] = np.nan                                              # one node: nothing to interpolate on
    d_sim, d_af = dev.to_device(sim), dev.to_device(af)
    for extrap in ("constant", "nan"):
        trace = dev.start_trace()
        got = K.qdm_adjust(dev, d_sim, d_af, q, kind, "nearest", extrap).get()
        dev.stop_trace()
        exp = osdba.qdm_adjust(sim, af, q, kind, "nearest", extrap)
        with np.errstate(invalid="ignore"):
            bad = np.argwhere(~(np.isclose(got, exp, rtol=1e-6, atol=0) | (np.isnan(got) & np.isnan(exp))))
        np.testing.assert_allclose(got, exp, rtol=1e-6, atol=0, equal_nan=True, err_msg=f"{extrap}: first mismatches (row, column) {bad[:5].tolist()}")
        with monkeypatch.context() as m:
            m.setenv("XH_DIAGNOSTICS", "1")
            m.setenv("XH_QDM_NOREGSORT", "1")
            legacy = K.qdm_adjust(dev, d_sim, d_af, q, kind, "nearest", extrap).get()
        np.testing.assert_array_equal(got, legacy)
    assert np.isnan(got[:, [52, 53, 54, 60, 61]]).all()


def test_qdm_precipitation_and_edge_cases(dev, rng):
    """Dry days (most samples tied at the minimum, rank 0), an all-NaN cell, a constant cell (0 / 0 ranks -> NaN), NaN
    factors (dropped nodes), the time-minor layout, argument errors."""
    from xclim_amd import kernels as K
    from xclim_amd import sdba as xsdba

    T, C = 730, 64
    def pr():
        x = rng.gamma(0.7, 4.0, (T, C)).astype(np.float32)
        x[rng.random((T, C)) < 0.6] = 0.0
        return x
    ref, hist, sim = pr(), pr(), pr()
    sim[:, 1] = np.nan
    sim[:, 2] = 3.0
    sim[5:40, 3] = np.nan
    z = np.flatnonzero(sim[:, 5] == 0.0)
    sim[z[::2], 5] = -0.0            # signed zeros tie with +0.0 (rankdata), they are not a smaller value
    qdm = xsdba.QuantileDeltaMapping.train(ref, hist, nquantiles=15, kind="*", device=dev)   # 0 / 0 factors at the dry nodes
    for interp in ("nearest", "linear"):
        got = qdm.adjust(sim, interp=interp)
        exp = osdba.qdm_adjust(sim, qdm.af, qdm.quantiles, "*", interp, "constant")
        np.testing.assert_allclose(got, exp, rtol=1e-6, equal_nan=True, err_msg=interp)
    assert np.isnan(got[:, 1]).all() and np.isnan(got[:, 2]).all()
    # time-minor (cells, time) layout gives the transposed result
    af = dev.to_device(qdm.af.reshape(15, C))
    tm = K.qdm_adjust(dev, dev.to_device(np.ascontiguousarray(sim.T)), af, qdm.quantiles, "*", "linear", time_axis=1).get()
    np.testing.assert_array_equal(tm.T, got)
    # cubic: a non-finite factor (x / 0: a wet reference quantile over a dry model quantile) among the nodes makes scipy's banded
    # spline solve spread inf / NaN in a pattern that depends on LAPACK's elimination order and on the dtype it is handed
    # (float32 factors here: some intervals stay finite; float64: everything inside the node range is NaN).  The kernel's rule
    # is the latter — NaN inside the node range, the end factors outside; such cells are compared on that rule only
    got_c = qdm.adjust(sim, interp="cubic")
    exp_c = osdba.qdm_adjust(sim, qdm.af, qdm.quantiles, "*", "cubic", "constant")
    bad = np.isinf(qdm.af).any(axis=0)
    assert bad.any() and not bad.all()
    # (the rank reaches the spline as a float32: the factor moves by 1e-7 of the NODE scale — where the spline swings through
    #  zero between a factor of 66 and one of 2.4 that is 4e-4 of the value itself: the tolerance is relative to the largest
    #  finite factor of the cell)
    fin = np.where(np.isfinite(qdm.af), np.abs(qdm.af), 0.0).max(axis=0)
    tol = 2e-6 * np.abs(exp_c) + 2e-6 * np.abs(np.nan_to_num(sim)) * fin[None, :]
    assert np.array_equal(np.isnan(got_c[:, ~bad]), np.isnan(exp_c[:, ~bad]))
    assert (np.abs(got_c - exp_c) <= tol)[:, ~bad][~np.isnan(exp_c[:, ~bad])].all()
    wet = ~np.isnan(sim[:, bad]) & (sim[:, bad] != 0)
    assert np.isnan(got_c[:, bad][wet]).mean() > 0.5      # (inside the node range; the driest wet days sit below the first valid node)
    with pytest.raises(ValueError):
        K.qdm_adjust(dev, dev.to_device(sim), af, qdm.quantiles[::-1].copy())


def test_qdm_preserves_quantile_deltas(dev, rng):
    """The defining property (Cannon et al. 2015): with additive factors the change between hist and sim of every quantile
    is carried over to the adjusted series: q(scen) - q(ref) ~ q(sim) - q(hist)."""
    from xclim_amd import sdba as xsdba

    T = 6000
    ref = rng.normal(10, 2, (T, 4)).astype(np.float32)
    hist = rng.normal(12, 3, (T, 4)).astype(np.float32)
    sim = (rng.normal(12, 3, (T, 4)) * 1.2 + 2.0).astype(np.float32)   # warmer and more variable future
    qdm = xsdba.QuantileDeltaMapping.train(ref, hist, nquantiles=50, kind="+", device=dev)
    scen = qdm.adjust(sim, interp="linear")
    qs = [0.1, 0.25, 0.5, 0.75, 0.9]
    d_model = np.quantile(sim, qs, axis=0) - np.quantile(hist, qs, axis=0)
    d_scen = np.quantile(scen, qs, axis=0) - np.quantile(ref, qs, axis=0)
    np.testing.assert_allclose(d_scen, d_model, atol=0.25)


@pytest.mark.parametrize("group,window", [("time.month", 1), ("time.dayofyear", 15)])
def test_qdm_grouped_matches_oracle(dev, rng, group, window):
    """QDM with a sub-grouping: trained like the grouped EQM, ranks taken inside each group's own time steps, factors of
    the step's group (oracle restatement; parity unpinned)."""
    from xclim_amd import sdba as xsdba

    T = 365 * 3
    ta, ot = _axes("2001-01-01", T, "noleap")
    shape = (T, 3, 5)
    ref = rng.normal(10, 3, shape).astype(np.float32)
    hist = rng.normal(11, 4, shape).astype(np.float32)
    sim = rng.normal(13, 4, shape).astype(np.float32)
    sim[rng.random(shape) < 0.02] = np.nan
    qdm = xsdba.QuantileDeltaMapping.train(ref, hist, nquantiles=10, kind="+", group=group, window=window, time=ta, device=dev)
    prop = group.split(".")[1]
    got = qdm.adjust(sim, interp="nearest", time=ta)
    exp = osdba.qdm_adjust_grouped(sim, ot, prop, qdm.group_labels, qdm.af, qdm.quantiles, "+", "nearest", "constant")
    np.testing.assert_allclose(got, exp, rtol=1e-6, equal_nan=True, err_msg=f"{group}")
    # interp="linear": xsdba interpolates the factors over the (quantile, group) plane with the quantile nodes themselves as
    # abscissa — a regular grid.  Day of year: every query lies on its group's row, where the answer is unique (and equals
    # the 1-D interpolation inside the group).  Month (fractional coordinate): every grid cell is a cocircular quadruple,
    # its Delaunay diagonal is Qhull's arbitrary choice — ours may take the other one: bounded by the cell's twist
    got = qdm.adjust(sim, interp="linear", time=ta)
    exp = osdba.qdm_adjust_grouped(sim, ot, prop, qdm.group_labels, qdm.af, qdm.quantiles, "+", "linear", "constant", mode="griddata")
    if prop == "dayofyear":
        np.testing.assert_allclose(got, exp, rtol=1e-6, atol=1e-6, equal_nan=True)
    else:
        af = qdm.af
        ext = np.concatenate([af[-1:], af, af[:1]])                      # cyclic rows 0 .. G + 1
        twist = np.abs(ext[1:, 1:] + ext[:-1, :-1] - ext[1:, :-1] - ext[:-1, 1:]).max(axis=(0, 1))   # per cell: max |mixed difference|
        assert np.array_equal(np.isnan(got), np.isnan(exp))
        assert (np.abs(got - exp) <= 0.5 * twist[None] + 1e-5)[~np.isnan(exp)].all()
        on_row = np.isclose(xsdba.Grouper(group).coordinate(ta, True) % 1.0, 0.0)      # the 15th / 16th of a month: on the row
        if on_row.any():
            np.testing.assert_allclose(got[on_row], exp[on_row], rtol=1e-6, atol=1e-6, equal_nan=True)
    with pytest.raises(NotImplementedError):
        qdm.adjust(sim, interp="cubic", time=ta)


@pytest.mark.parametrize("kind", ["+", "*"])
@pytest.mark.parametrize("detrend", [0, 1])
def test_dqm_matches_oracle(dev, rng, kind, detrend):
    """DetrendedQuantileMapping: train (mean-normalised quantiles + scaling) and adjust (scale, detrend, EQM lookup, retrend)
    against the oracle restatement; NaN steps, an all-NaN cell.  Parity unpinned (xsdba absent)."""
    from xclim_amd import sdba as xsdba

    T, cells = 1460, (6, 7)
    t = np.arange(T, dtype=np.float32)[:, None, None]
    ref = (rng.normal(10, 3, (T,) + cells) + 0.001 * t).astype(np.float32)
    hist = (rng.normal(12, 4, (T,) + cells) + 0.001 * t).astype(np.float32)
    sim = (rng.normal(14, 4, (T,) + cells) + 0.004 * t).astype(np.float32)   # a stronger trend in the future run
    if kind == "*":
        ref, hist, sim = np.abs(ref) + 1, np.abs(hist) + 1, np.abs(sim) + 1
    sim[rng.random(sim.shape) < 0.02] = np.nan
    ref[rng.random(ref.shape) < 0.01] = np.nan
    sim[:, 0, 1] = np.nan
    dqm = xsdba.DetrendedQuantileMapping.train(ref, hist, nquantiles=20, kind=kind, device=dev)
    eaf, ehq, esc = osdba.dqm_train(ref, hist, 20, kind)
    np.testing.assert_allclose(dqm.scaling, esc.reshape(cells), rtol=1e-12)
    np.testing.assert_allclose(dqm.hist_q, ehq, rtol=1e-6, equal_nan=True)
    np.testing.assert_allclose(dqm.af, eaf, rtol=2e-5, atol=2e-6, equal_nan=True)   # differences of nearly equal quantiles
    for interp in ("nearest", "linear"):
        got = dqm.adjust(sim, interp=interp, detrend=detrend)
        # the oracle adjusts with the DEVICE's factors: the node lookup is discontinuous in them for "nearest"
        exp = osdba.dqm_adjust(sim, dqm.af, dqm.hist_q, dqm.scaling, kind, interp, "constant", detrend)
        np.testing.assert_allclose(got, exp, rtol=2e-6, atol=1e-6, equal_nan=True, err_msg=f"{interp}")
    assert np.isnan(got[:, 0, 1]).all()
    with pytest.raises(NotImplementedError):
        dqm.adjust(sim, detrend=2)


def test_dqm_removes_the_bias_and_keeps_the_trend(dev, rng):
    """What DQM is for: the adjusted series has the distribution of ref around the (scaled) trend of sim — the linear trend
    of sim survives the adjustment, the mean bias of the model does not."""
    from xclim_amd import sdba as xsdba

    T = 7300
    t = np.arange(T, dtype=np.float64)[:, None]
    ref = rng.normal(10, 2, (T, 3))
    hist = rng.normal(13, 3, (T, 3))                      # model: +3 bias, too variable
    sim = rng.normal(13, 3, (T, 3)) + 0.0005 * t         # the same model with a trend of +3.65 over the run
    dqm = xsdba.DetrendedQuantileMapping.train(ref.astype(np.float32), hist.astype(np.float32), nquantiles=50, kind="+", device=dev)
    scen = dqm.adjust(sim.astype(np.float32), interp="linear", detrend=1)
    slope = np.polyfit(t[:, 0], scen, 1)[0]
    np.testing.assert_allclose(slope, 0.0005, rtol=0.1)
    np.testing.assert_allclose(scen.mean(axis=0), 10.0 + 0.0005 * (T - 1) / 2, atol=0.15)
    np.testing.assert_allclose((scen - np.polyval(np.polyfit(t[:, 0], scen, 1), t)).std(axis=0), 2.0, rtol=0.05)


@pytest.mark.parametrize("freq", ["W", "W-THU", "7D", "10D"])
def test_weekly_and_nday_resampling(dev, rng, freq):
    """threshold_count / select_resample_op / run statistics with weekly and n-day bins (pandas "W" semantics: closed and
    labelled on the right) against the oracle, which resamples with pandas itself."""
    T = 400
    ta, ot = _axes("2003-12-27", T)
    x = rng.normal(285, 6, (T, 5, 7)).astype(np.float32)
    x[rng.random(x.shape) < 0.02] = np.nan
    np.testing.assert_array_equal(xgen.threshold_count(x, ">", 288.0, ta, freq, device=dev), ogen.threshold_count(x, ">", 288.0, ot, freq))
    np.testing.assert_allclose(xgen.select_resample_op(x, "mean", ta, freq, device=dev), ogen.select_resample_op(x, "mean", ot, freq),
                               rtol=1e-6, equal_nan=True)
    got = xgen.spell_length_statistics(x, 284.0, 1, None, ">", "max", ta, freq, device=dev)
    np.testing.assert_array_equal(got, ogen.spell_length_statistics(x, 284.0, 1, None, ">", "max", ot, freq))


@pytest.mark.parametrize("include", [True, (False, True), (True, False)])
def test_select_time_with_per_cell_doy_bounds(dev, rng, include):
    """select_time(doy_bounds=(start, end)) with array bounds (mask_between_doys, cal:1199-1257, no time dimension on the
    bounds): spans inside the year, spans that cross the new year (start > end), an int against an array, NaN = open
    bound; drop=True is refused like in the reference."""
    from xclim_amd.calendar import select_time

    T, cells = 800, (6, 8)
    ta, ot = _axes("2001-01-01", T)
    x = rng.normal(280, 5, (T,) + cells).astype(np.float32)
    start = rng.integers(1, 366, cells).astype(np.float64)
    end = rng.integers(1, 366, cells).astype(np.float64)
    start[0, 0], end[0, 1] = np.nan, np.nan
    for bounds in ((start, end), (100, end), (start, 250)):
        got = select_time(x, ta, doy_bounds=bounds, include_bounds=include, device=dev)
        exp = ocal.select_time(x, ot, doy_bounds=bounds, include_bounds=include)
        np.testing.assert_array_equal(got, exp)
    assert np.isnan(got).any() and (~np.isnan(got)).any()
    with pytest.raises(ValueError, match="incompatible with drop=True"):
        select_time(x, ta, drop=True, doy_bounds=(start, end), device=dev)


@pytest.mark.parametrize("include", [True, (False, True)])
@pytest.mark.parametrize("start_date,cal,freq", [("2000-03-15", "standard", "YS"), ("2001-01-01", "noleap", "YS-JUL"), ("2001-01-01", "standard", "MS")])
def test_select_time_with_doy_bounds_that_carry_a_time_dimension(dev, rng, include, start_date, cal, freq):
    """mask_between_doys with bounds (period, *cells) (core/calendar.py:1211-1246): one (start, end) pair per period of the
    bounds' own frequency and per cell, compared as DAYS SINCE the period label (doy_to_days_since: a doy below the
    label's doy lies in the next year); NaN = open, periods the bounds do not label are masked entirely, a first
    period that starts before the data, a calendar without leap days, July-anchored years, monthly bounds."""
    from xclim_amd.calendar import select_time

    T, cells = 800, (5, 8)
    ta, ot = _axes(start_date, T) if cal == "standard" else _axes(start_date, T, cal)
    x = rng.normal(280, 5, (T,) + cells).astype(np.float32)
    seg, starts = ta.segments(freq)
    labels = [(y, m, 1) for (y, m) in starts][:-1]              # the last period of the data has no bounds
    nb = len(labels)
    if freq == "MS":
        base = np.array([TimeAxis(np.array([y]), np.array([m]), np.array([1]), ta.calendar).doy[0] for y, m, _ in labels])
        start = base[:, None, None] + rng.integers(0, 12, (nb,) + cells).astype(np.float64)
        end = start + rng.integers(0, 20, (nb,) + cells)
    else:
        start = rng.integers(1, 366, (nb,) + cells).astype(np.float64)
        end = rng.integers(1, 366, (nb,) + cells).astype(np.float64)
    start[0, 0, 0], end[0, 0, 1] = np.nan, np.nan
    bt = TimeAxis(np.array([l[0] for l in labels]), np.array([l[1] for l in labels]), np.array([1] * nb), ta.calendar)
    got = select_time(x, ta, doy_bounds=(start, end), include_bounds=include, bounds_time=bt, device=dev)
    exp = ocal.select_time(x, ot, doy_bounds=(start, end), include_bounds=include, bounds_labels=labels, bounds_freq=freq)
    np.testing.assert_array_equal(got, exp)
    assert np.isnan(got[int(seg[-2]):]).all() and (~np.isnan(got)).any()     # the unlabelled last period is masked
    got2 = select_time(x, ta, doy_bounds=(start, end), include_bounds=include, bounds_time=bt, bounds_freq=freq, device=dev)
    np.testing.assert_array_equal(got2, exp)


@pytest.mark.parametrize("calendar", ["standard", "noleap", "360_day"])
def test_missing_methods_match_oracle(dev, rng, calendar):
    """core/missing.py:325-512 (VERDICT r4 missing #7): MissingSomeButNotAll, MissingWMO, MissingPct, AtLeastNValid — and
    MissingAny with freq=None — from the valid counts of xh_resample_reduce and, for WMO, the longest run of NaN steps per month
    (xh_run_stats on the x != x mask); bit-exact (boolean) against the oracle's group loops, with and without a time
    selection, one- and two-step (subfreq="MS")."""
    from oracle import missing as omiss
    from xclim_amd import missing as hmiss

    T = 800
    if calendar == "standard":
        ta, ot = TimeAxis.daily("2000-03-15", T, "standard"), OTime.standard("2000-03-15", T)
    else:
        ta = TimeAxis.daily("2000-01-01", T + 74, calendar).subset(slice(74, None))
        ot = OTime.noleap(2000, T + 74, calendar).isel(slice(74, None))
    x = rng.normal(280, 5, (T, 6, 9)).astype(np.float32)
    x[rng.random(x.shape) < 0.04] = np.nan                  # scattered missing days
    for c in range(9):                                       # runs of 2 .. 12 missing days, whole missing months, an empty cell
        t0 = int(rng.integers(0, T - 40))
        x[t0:t0 + 2 + c + (c > 5) * 3, c % 6, c] = np.nan
    x[100:170, 2, 3] = np.nan
    x[:, 5, 8] = np.nan
    x[:, 0, 0] = 281.0                                        # a complete cell
    for freq in ("MS", "YS", "QS-DEC", "YS-JUL"):
        for ix in ({}, {"month": [1, 6, 7]}, {"season": "JJA"}):
            np.testing.assert_array_equal(hmiss.missing_wmo(x, freq, ta, device=dev, **ix), omiss.missing_wmo(x, ot, freq, **ix), err_msg=f"wmo {freq} {ix}")
            np.testing.assert_array_equal(hmiss.missing_wmo(x, freq, ta, nm=5, nc=3, device=dev, **ix), omiss.missing_wmo(x, ot, freq, 5, 3, **ix))
            for sub in (None, "MS"):
                np.testing.assert_array_equal(hmiss.missing_pct(x, freq, ta, 0.1, sub, device=dev, **ix), omiss.missing_pct(x, ot, freq, 0.1, sub, **ix),
                                              err_msg=f"pct {freq} {sub} {ix}")
                np.testing.assert_array_equal(hmiss.at_least_n_valid(x, freq, ta, 20, sub, device=dev, **ix), omiss.at_least_n_valid(x, ot, freq, 20, sub, **ix),
                                              err_msg=f"n {freq} {sub} {ix}")
            np.testing.assert_array_equal(hmiss.missing_some_but_not_all(x, freq, ta, device=dev, **ix), omiss.missing_some_but_not_all(x, ot, freq, **ix))
            np.testing.assert_array_equal(hmiss.missing_any(x, freq, ta, device=dev, **ix), omiss.missing_any(x, ot, freq, **ix))
        np.testing.assert_array_equal(hmiss.missing_pct(x, freq, ta, 0.02, device=dev, doy_bounds=(150, 250)), omiss.missing_pct(x, ot, freq, 0.02, doy_bounds=(150, 250)))
    for ix in ({}, {"month": [7]}):                           # freq=None: the whole series is one period (core/missing.py:124-127)
        np.testing.assert_array_equal(hmiss.missing_any(x, None, ta, device=dev, **ix), omiss.missing_any(x, ot, None, **ix))
        np.testing.assert_array_equal(hmiss.missing_pct(x, None, ta, 0.05, device=dev, **ix), omiss.missing_pct(x, ot, None, 0.05, **ix))
        np.testing.assert_array_equal(hmiss.missing_pct(x, None, ta, 0.3, "MS", device=dev, **ix), omiss.missing_pct(x, ot, None, 0.3, "MS", **ix))
        np.testing.assert_array_equal(hmiss.missing_wmo(x, None, ta, device=dev, **ix), omiss.missing_wmo(x, ot, None, **ix))
    assert hmiss.missing_wmo(x, "YS", ta, device=dev).any() and not hmiss.missing_wmo(x, "YS", ta, device=dev)[1, 0, 0]
    with pytest.raises(ValueError, match="not valid for MissingWMO"):
        hmiss.missing_wmo(x, "MS", ta, nm=31, device=dev)
    with pytest.raises(ValueError, match="not valid for MissingPct"):
        hmiss.missing_pct(x, "MS", ta, tolerance=1.5, device=dev)


def test_reference_missing_answers_on_the_device(dev):
    """/root/reference/tests/test_missing.py:166-285 through the HIP path (the same answers the oracle is pinned to in
    tests/test_oracle_reference_answers.py)."""
    from xclim_amd import missing as hmiss

    t = lambda n: TimeAxis.daily("2000-07-01", n, "standard")  # noqa: E731
    a = np.arange(360.0, dtype=np.float32)
    a[5:7] = np.nan
    a[40:45] = np.nan
    a[70:92:2] = np.nan
    out = hmiss.missing_wmo(a, "MS", t(360), device=dev)
    assert not out[0] and out[1] and out[2]
    a = np.arange(350.0, dtype=np.float32)
    a[5:16] = np.nan
    np.testing.assert_array_equal(hmiss.missing_wmo(a, "QS-JAN", t(350), device=dev), [True, False, False, True])
    np.testing.assert_array_equal(hmiss.missing_wmo(np.arange(31.0, dtype=np.float32), "YS", t(31), device=dev), [True])
    a = np.arange(360.0, dtype=np.float32)
    a[5:7] = np.nan
    a[40:45] = np.nan
    out = hmiss.missing_pct(a, "MS", t(360), tolerance=0.1, device=dev)
    assert not out[0] and out[1]
    a = np.arange(360.0, dtype=np.float32)
    a[5:10] = np.nan
    a[40:55] = np.nan
    np.testing.assert_array_equal(hmiss.at_least_n_valid(a, "MS", t(360), n=20, device=dev)[:2], [False, True])
    a = np.arange(360.0, dtype=np.float32)
    a[:40] = np.nan
    out = hmiss.missing_some_but_not_all(a, "MS", t(360), device=dev)
    assert not out[0] and out[1] and not out[2]
    np.testing.assert_array_equal(hmiss.missing_any(np.zeros(360, np.float32), None, t(360), device=dev), [False])
    np.testing.assert_array_equal(hmiss.missing_any(np.zeros(360, np.float32), None, t(360), device=dev, month=[7]), [False])


@pytest.mark.parametrize("years,window,nq,kind", [(4, 5, 7, "+"), (6, 9, 20, "*"), (3, 31, 12, "+"), (30, 31, 20, "+")])
def test_eqm_doy_window_sliding_matches_per_group(dev, rng, monkeypatch, years, window, nq, kind):
    """winsel.hip (round 6): day-of-year groups with a window — every cell keeps its window SORTED from one day to the next
    (xh_eqm_train_window) instead of selecting each of the 365 groups from its gathered sample (xh_eqm_train per group: the
    route of rounds 2-5, kept behind XH_WINSEL=0).  BIT-IDENTICAL tables: NaN samples, ties (rounded values), infinities, an
    empty cell, windows that reach beyond both ends of the series, 30 years x 31 days = 930 samples."""
    from xclim_amd import sdba as xsdba

    T = 365 * years
    cells = 29 if years < 30 else 40
    ta = TimeAxis.daily("2001-01-01", T, "noleap")
    t = np.arange(T)[:, None]
    ref = (288 + 10 * np.sin(2 * np.pi * t / 365) + rng.normal(0, 3, (T, cells))).astype(np.float32)
    hist = (ref[::-1] * 1.01 + rng.normal(0, 1, (T, cells))).astype(np.float32)
    ref[rng.random(ref.shape) < 0.03] = np.nan
    hist[rng.random(hist.shape) < 0.02] = np.nan
    if years == 6:
        ref = np.round(ref, 1)        # ties: many equal samples leave and enter
        hist = np.round(hist, 0)
    hist[:, 3] = np.nan                # a cell without samples
    ref[40:50, 5] = np.inf
    hist[100:103, 6] = -np.inf
    monkeypatch.setenv("XH_DIAGNOSTICS", "1")
    monkeypatch.setenv("XH_WINSEL", "1")
    trace = dev.start_trace()
    a = xsdba.EmpiricalQuantileMapping.train(ref, hist, nquantiles=nq, kind=kind, group="time.dayofyear", window=window, time=ta, device=dev)
    dev.stop_trace()
    assert [n for n, _ in trace].count("xh_eqm_train_window") == 1 and "xh_eqm_train" not in [n for n, _ in trace]
    monkeypatch.setenv("XH_WINSEL", "0")
    trace = dev.start_trace()
    b = xsdba.EmpiricalQuantileMapping.train(ref, hist, nquantiles=nq, kind=kind, group="time.dayofyear", window=window, time=ta, device=dev)
    dev.stop_trace()
    assert [n for n, _ in trace].count("xh_eqm_train") == 365
    np.testing.assert_array_equal(a.hist_q, b.hist_q)
    np.testing.assert_array_equal(a.af, b.af)
    assert np.isnan(a.hist_q[:, :, 3]).all() and np.isfinite(a.hist_q[:, :, 0]).all()


@pytest.mark.parametrize("years,window", [(6, 7), (30, 31)])
def test_eqm_doy_window_sliding_standard_calendar(dev, rng, monkeypatch, years, window):
    """The same on a calendar WITH leap days (Grouper.sliding_stretches: the schedule from the groups' sample rows — what leaves
    and enters between consecutive days of the year, up to 64 rows each way): with 30 years and a window of 31 days day 366 (only
    the leap years have it: 682 rows would leave at once) is not part of the stretch and is selected from its gathered sample;
    bit-identical to the per-group path."""
    from xclim_amd import sdba as xsdba

    T = int(365.25 * years) + 100
    ta = TimeAxis.daily("2000-01-01", T, "standard")
    cells, nq = 23, 10
    t = np.arange(T)[:, None]
    ref = np.round(288 + 10 * np.sin(2 * np.pi * t / 365.25) + rng.normal(0, 3, (T, cells)), 1).astype(np.float32)
    hist = (ref[::-1] * 1.01 + rng.normal(0, 1, (T, cells))).astype(np.float32)
    ref[rng.random(ref.shape) < 0.03] = np.nan
    hist[:, 2] = np.nan
    stretches, rest, _ = xsdba.Grouper("time.dayofyear", window).sliding_stretches(ta)
    assert len(stretches) == 1 and stretches[0][0] == 0 and rest == ([] if years == 6 else [365])
    monkeypatch.setenv("XH_DIAGNOSTICS", "1")
    monkeypatch.setenv("XH_WINSEL", "1")
    trace = dev.start_trace()
    a = xsdba.EmpiricalQuantileMapping.train(ref, hist, nquantiles=nq, kind="+", group="time.dayofyear", window=window, time=ta, device=dev)
    dev.stop_trace()
    names = [n for n, _ in trace]
    assert names.count("xh_eqm_train_window") == 1 and names.count("xh_eqm_train") == len(rest)
    monkeypatch.setenv("XH_WINSEL", "0")
    b = xsdba.EmpiricalQuantileMapping.train(ref, hist, nquantiles=nq, kind="+", group="time.dayofyear", window=window, time=ta, device=dev)
    assert a.hist_q.shape == (366, nq, cells)
    np.testing.assert_array_equal(a.hist_q, b.hist_q)
    np.testing.assert_array_equal(a.af, b.af)


@pytest.mark.parametrize("years,window,nq,kind,cal", [(4, 5, 7, "+", "noleap"), (6, 9, 20, "*", "noleap"), (30, 31, 20, "+", "noleap"),
                                                      (30, 31, 15, "*", "standard")])
def test_dqm_doy_window_sliding_matches_per_group(dev, rng, monkeypatch, years, window, nq, kind, cal):
    """xh_dqm_train_window (round 6): the training of a DETRENDED quantile mapping with day-of-year groups and a window through the
    sorted sliding window — the mean of every group's sample from the window, the picked samples normalised — against the
    per-group chain it replaces (XH_WINSEL=0: gather, xh_poly_trend degree 0, xh_trend_apply, xh_eqm_train per group).  The means
    differ in their summation order only (1e-14); the tables are BIT-IDENTICAL wherever the two means are, and within 1e-6
    elsewhere.  Cells: NaN samples, ties, infinities in the window, a negative mean (kind "*": the order reverses), all zeros
    (0 / 0), no valid sample; a standard calendar (day 366 from its gathered sample)."""
    from xclim_amd import sdba as xsdba

    T = 365 * years + (years // 4 + 60 if cal == "standard" else 0)
    cells = 29 if years < 30 else 40
    ta = TimeAxis.daily("2000-01-01", T, cal)
    t = np.arange(T)[:, None]
    ref = (28 + 10 * np.sin(2 * np.pi * t / 365) + rng.normal(0, 3, (T, cells))).astype(np.float32)
    hist = (ref[::-1] * 1.01 + rng.normal(0, 1, (T, cells))).astype(np.float32)
    ref[rng.random(ref.shape) < 0.03] = np.nan
    hist[rng.random(hist.shape) < 0.02] = np.nan
    if years == 6:
        ref = np.round(ref, 1)
        hist = np.round(hist, 0)
    hist[:, 3] = np.nan
    ref[40:50, 5] = np.inf
    hist[100:103, 6] = -np.inf
    ref[:, 7] = -ref[:, 7]
    hist[:, 8] = 0.0
    hist[300:330, 9] = 1e38       # (kind "*": x / mean overflows for none, the mean stays finite; "+": nothing special)
    monkeypatch.setenv("XH_DIAGNOSTICS", "1")
    monkeypatch.setenv("XH_WINSEL", "1")
    trace = dev.start_trace()
    a = xsdba.DetrendedQuantileMapping.train(ref, hist, nquantiles=nq, kind=kind, group="time.dayofyear", window=window, time=ta, device=dev)
    dev.stop_trace()
    names = [n for n, _ in trace]
    G = 365 if cal == "noleap" else 366
    rest = 1 if (cal == "standard" and window > 1) else 0   # (day 366 with a window: 682 rows would leave at once)
    assert names.count("xh_dqm_train_window") == 1 and names.count("xh_eqm_train") == rest
    monkeypatch.setenv("XH_WINSEL", "0")
    trace = dev.start_trace()
    b = xsdba.DetrendedQuantileMapping.train(ref, hist, nquantiles=nq, kind=kind, group="time.dayofyear", window=window, time=ta, device=dev)
    dev.stop_trace()
    assert [n for n, _ in trace].count("xh_eqm_train") == G
    sa, sb = a.scaling, b.scaling
    assert sa.shape == (G, cells)
    np.testing.assert_allclose(sa, sb, rtol=1e-13, equal_nan=True)
    same = (sa == sb) | (np.isnan(sa) & np.isnan(sb))          # (scaling = mean(ref) OP mean(hist): equal bits <- equal means, nearly always)
    assert same.mean() > 0.5
    ha, hb = a.hist_q, b.hist_q
    fa, fb = a.af, b.af
    m = np.broadcast_to(same[:, None, :], ha.shape)
    np.testing.assert_array_equal(ha[m], hb[m])
    np.testing.assert_array_equal(fa[m], fb[m])
    with np.errstate(all="ignore"):
        np.testing.assert_allclose(ha, hb, rtol=1e-6, atol=1e-6)
        ok = np.isfinite(fb) & (np.abs(hb) > 1e-3)
        np.testing.assert_allclose(fa[ok], fb[ok], rtol=2e-5, atol=1e-5)
    assert np.isnan(ha[:, :, 3]).all() and np.isfinite(ha[:, :, 0]).all()
    if kind == "*":
        assert np.isnan(ha[:, :, 8]).all()


@pytest.mark.parametrize("C", [37, 256])
def test_dqm_groups_kernels_match_per_group_calls(dev, rng, C):
    """xh_poly_trend_groups / xh_trend_apply_groups (round 6: the per-group fit and the per-group trend of a grouped
    DetrendedQuantileMapping.adjust in ONE launch each, a group = a list of rows) against xh_poly_trend_u / xh_trend_apply_u on
    every group's gathered rows — bit for bit: groups of unequal size, an empty group, rows in no group (left untouched), NaN
    samples, a cell without samples, degree 0 and 1, all four operations, in place."""
    from xclim_amd import kernels as K

    T, G = 400, 9
    x = rng.normal(10, 4, (T, C)).astype(np.float32)
    x[rng.random(x.shape) < 0.1] = np.nan
    x[:, 2] = np.nan
    gid = rng.integers(-1, G, T)
    gid[gid == 4] = 5                  # group 4 is empty
    perm = np.argsort(gid, kind="stable")
    perm = perm[gid[perm] >= 0]
    counts = np.bincount(gid[gid >= 0], minlength=G)
    offs = np.concatenate([[0], np.cumsum(counts)])
    u = rng.normal(0, 50, T)
    d_x, d_u = dev.to_device(x), dev.to_device(u, dtype=np.float64)
    for degree in (0, 1):
        p0, p1 = K.poly_trend_groups(dev, d_x, perm, offs, d_u, degree)
        P0, P1 = p0.get(), (p1.get() if p1 is not None else None)
        for op in "+-*/":
            out = dev.to_device(np.full((T, C), -7.0, np.float32))
            K.trend_apply_groups(dev, d_x, perm, offs, p0, p1, op, u=d_u, out=out)
            got = out.get()
            assert (got[gid < 0] == -7.0).all()
            for g in range(G):
                rows = perm[offs[g]:offs[g + 1]]
                if not len(rows):
                    assert np.isnan(P0[g]).all()
                    continue
                blk = K.select_rows(dev, d_x, rows)
                ug = dev.to_device(np.ascontiguousarray(u[rows]), dtype=np.float64)
                q0, q1 = K.poly_trend(dev, blk, degree, u=ug)
                np.testing.assert_array_equal(P0[g], q0.get())
                if degree:
                    np.testing.assert_array_equal(P1[g], q1.get())
                np.testing.assert_array_equal(got[rows], K.trend_apply(dev, blk, q0, q1, op, u=ug).get())
        # in place, and a per-group constant without a coordinate
        y = dev.to_device(x)
        K.trend_apply_groups(dev, y, perm, offs, p0, None, "-", out=y)
        exp = x.astype(np.float64) - P0[np.maximum(gid, 0)]
        keep = gid >= 0
        np.testing.assert_array_equal(y.get()[keep], exp.astype(np.float32)[keep])
        np.testing.assert_array_equal(y.get()[~keep], x[~keep])


@pytest.mark.parametrize("years,kind,interp,extrap", [(5, "+", "nearest", "constant"), (30, "*", "linear", "constant"),
                                                      (40, "+", "linear", "constant"), (64, "*", "nearest", "nan")])
def test_qdm_small_groups_match_per_group_calls(dev, rng, monkeypatch, years, kind, interp, extrap):
    """xh_qdm_adjust_groups (round 6): QuantileDeltaMapping.adjust with a day-of-year grouping — every step ranked among the steps
    of its own group, all 365 groups in ONE launch (keys in registers, <= 64 rows per group) — against the loop it replaces
    (XH_QDM_GROUPS=0: a gather and xh_qdm_adjust per group), BIT FOR BIT: NaN samples, ties (dry days: zeros tie, -0.0 with 0.0),
    constant groups, a cell without samples, NaN factors (dropped nodes), a cell with fewer than two valid nodes, 5 to 64 years
    (both register sizes), day 366 of a standard calendar (a smaller group)."""
    from xclim_amd import sdba as xsdba

    cells, nq = 37, 12
    T = 365 * years + years // 4
    ta = TimeAxis.daily("2000-01-01", T, "standard")
    t = np.arange(T)[:, None]
    ref = (10 + 8 * np.sin(2 * np.pi * t / 365.25) + rng.normal(0, 3, (T, cells))).astype(np.float32) + (20 if kind == "*" else 0)
    hist = (ref * 1.05 + rng.normal(0, 1, (T, cells))).astype(np.float32)
    sim = (hist + 2 + rng.normal(0, 1, (T, cells))).astype(np.float32)
    sim[rng.random(sim.shape) < 0.05] = np.nan
    sim[:, 1] = np.round(sim[:, 1])                                 # ties
    sim[:, 2] = np.where(rng.random(T) < 0.6, 0.0, sim[:, 2])       # dry days
    sim[::7, 2] = -0.0
    sim[:, 3] = np.nan
    sim[:, 4] = 5.0                                                 # every group constant: pct = 0 / 0
    hist[:, 5] = np.nan                                             # all factors NaN: no node
    m = xsdba.QuantileDeltaMapping.train(ref, hist, nquantiles=nq, kind=kind, group="time.dayofyear", window=1, time=ta, device=dev)
    af = m.af.copy()
    af[:, 3:6, 6] = np.nan                                          # dropped nodes
    af[:, 1:, 7] = np.nan                                           # one valid node: no interpolation
    m._af = dev.to_device(np.ascontiguousarray(af.reshape(af.shape[0], nq, cells)))
    monkeypatch.setenv("XH_DIAGNOSTICS", "1")
    monkeypatch.setenv("XH_QDM_GROUPS", "1")
    trace = dev.start_trace()
    a = m.adjust(sim, interp=interp, extrapolation=extrap, time=ta)
    dev.stop_trace()
    names = [n for n, _ in trace]
    assert names.count("xh_qdm_adjust_groups") == 1 and "xh_qdm_adjust" not in names
    monkeypatch.setenv("XH_QDM_GROUPS", "0")
    trace = dev.start_trace()
    b = m.adjust(sim, interp=interp, extrapolation=extrap, time=ta)
    dev.stop_trace()
    assert [n for n, _ in trace].count("xh_qdm_adjust") == 366
    np.testing.assert_array_equal(a, b)
    assert np.isnan(a[:, 3]).all() and np.isnan(a[:, 5]).all() and np.isnan(a[:, 7]).all()
    assert np.isfinite(a[:, 0]).mean() > (0.9 if extrap == "constant" else 0.7)   # ("nan": the ranks beyond the end nodes)


@pytest.mark.parametrize("years,nq,kind,cal", [(5, 9, "+", "noleap"), (30, 20, "*", "standard"), (33, 15, "+", "noleap"), (64, 7, "*", "noleap")])
def test_eqm_dqm_doy_training_without_window_in_one_launch(dev, rng, monkeypatch, years, nq, kind, cal):
    """xh_eqm_train_groups / xh_dqm_train_groups (round 6): group="time.dayofyear" WITHOUT a window — 365 groups of one row per
    year — trained in one launch per field (thread = one cell of one group: the rows as keys in registers, a bitonic network,
    quantiles by position; DQM: the group's mean in the rows' order and the samples normalised before the sort) against the
    per-group loop (XH_TRAIN_GROUPS=0), BIT FOR BIT — tables AND scaling: NaN samples, ties, infinities, a negative mean, an all-zero
    cell (0 / 0), a cell without samples, both register sizes, day 366 of a standard calendar (a smaller group)."""
    from xclim_amd import sdba as xsdba

    T = 365 * years + (years // 4 + 30 if cal == "standard" else 0)
    cells = 41
    ta = TimeAxis.daily("2000-01-01", T, cal)
    t = np.arange(T)[:, None]
    ref = (28 + 10 * np.sin(2 * np.pi * t / 365) + rng.normal(0, 3, (T, cells))).astype(np.float32)
    hist = (ref[::-1] * 1.01 + rng.normal(0, 1, (T, cells))).astype(np.float32)
    ref[rng.random(ref.shape) < 0.05] = np.nan
    hist[rng.random(hist.shape) < 0.02] = np.nan
    ref[:, 1] = np.round(ref[:, 1])
    hist[:, 3] = np.nan
    ref[40:50, 5] = np.inf
    hist[100:103, 6] = -np.inf
    ref[:, 7] = -ref[:, 7]
    hist[:, 8] = 0.0
    monkeypatch.setenv("XH_DIAGNOSTICS", "1")
    for Model, entry in ((xsdba.EmpiricalQuantileMapping, "xh_eqm_train_groups"), (xsdba.DetrendedQuantileMapping, "xh_dqm_train_groups")):
        monkeypatch.setenv("XH_TRAIN_GROUPS", "1")
        trace = dev.start_trace()
        a = Model.train(ref, hist, nquantiles=nq, kind=kind, group="time.dayofyear", time=ta, device=dev)
        dev.stop_trace()
        names = [n for n, _ in trace]
        assert names.count(entry) == 1 and "xh_eqm_train" not in names
        monkeypatch.setenv("XH_TRAIN_GROUPS", "0")
        trace = dev.start_trace()
        b = Model.train(ref, hist, nquantiles=nq, kind=kind, group="time.dayofyear", time=ta, device=dev)
        dev.stop_trace()
        G = 365 if cal == "noleap" else 366
        assert [n for n, _ in trace].count("xh_eqm_train") == G and a.hist_q.shape == (G, nq, cells)
        np.testing.assert_array_equal(a.hist_q, b.hist_q)
        np.testing.assert_array_equal(a.af, b.af)
        if Model is xsdba.DetrendedQuantileMapping:
            np.testing.assert_array_equal(a.scaling, b.scaling)
            if kind == "*":
                assert np.isnan(a.hist_q[:, :, 8]).all()
        assert np.isnan(a.hist_q[:, :, 3]).all() and np.isfinite(a.hist_q[:, :, 0]).all()


@pytest.mark.parametrize("kind", ["+", "*"])
@pytest.mark.parametrize("group", ["time.month", "time.season"])
def test_dqm_training_without_window_matches_per_group_chain(dev, rng, kind, group):
    """DetrendedQuantileMapping.train with a sub-grouping and no window (round 6): the group means are one xh_poly_trend_groups
    per field, the normalisation one xh_trend_apply_groups per field over the series where it lies, the tables the grouped EQM
    training of the normalised series — against dqm_train group by group on gathered rows (xh_poly_trend degree 0,
    xh_trend_apply, xh_eqm_train), BIT FOR BIT: tables and scaling; NaN samples, a cell without samples, a negative mean."""
    from xclim_amd import kernels as K
    from xclim_amd import sdba as xsdba

    T, cells, nq = 365 * 3, 9, 8
    ta = TimeAxis.daily("2001-01-01", T, "noleap")
    t = np.arange(T)[:, None]
    ref = (20 + 8 * np.sin(2 * np.pi * t / 365) + rng.normal(0, 3, (T, cells))).astype(np.float32)
    hist = (ref[::-1] * 1.05 + rng.normal(0, 1, (T, cells))).astype(np.float32)
    ref[rng.random(ref.shape) < 0.05] = np.nan
    hist[:, 2] = np.nan
    ref[:, 3] = -ref[:, 3]
    m = xsdba.DetrendedQuantileMapping.train(ref, hist, nquantiles=nq, kind=kind, group=group, time=ta, device=dev)
    grp = xsdba.Grouper(group)
    gi = grp.index(ta, m.group_labels)
    d_r, d_h = dev.to_device(ref), dev.to_device(hist)
    inv = "-" if kind == "+" else "/"
    for g in range(len(m.group_labels)):
        rows = np.nonzero(gi == g)[0]
        rg, hg = K.select_rows(dev, d_r, rows), K.select_rows(dev, d_h, rows)
        mu_r, _ = K.poly_trend(dev, rg, 0)
        mu_h, _ = K.poly_trend(dev, hg, 0)
        a_g, h_g = K.eqm_train(dev, K.trend_apply(dev, rg, mu_r, None, inv), K.trend_apply(dev, hg, mu_h, None, inv), m.quantiles, kind)
        np.testing.assert_array_equal(m.hist_q[g], h_g.get(), err_msg=f"group {g}")
        np.testing.assert_array_equal(m.af[g], a_g.get(), err_msg=f"group {g}")
        with np.errstate(all="ignore"):
            np.testing.assert_array_equal(m.scaling[g], mu_r.get() - mu_h.get() if kind == "+" else mu_r.get() / mu_h.get())


@pytest.mark.parametrize("T", [365, 930, 3000])
@pytest.mark.parametrize("kind", ["+", "*"])
def test_eqm_train_nan_nodes_masks_and_zero_over_zero(dev, rng, kind, T):
    """The NaN rule of the training tables (utl:552-554: a NaN node of a series with valid samples is its largest valid sample) as
    two passes since round 6 — pass 1 corrects the columns without a NaN node and lists the 64-column groups that hold one, pass 2
    scans the series of those columns — on a field with a land / sea mask (all-NaN columns keep their NaN), columns with
    infinities (inf - inf in the lerp: the node becomes the largest valid sample) and, in the SAME group, a column whose
    correction is itself NaN (0 / 0: not a node to repair, and not to be corrected twice), against the oracle."""
    C, nq = 200, 20
    ref = (10 + rng.normal(0, 3, (T, C))).astype(np.float32)
    hist = (11 + rng.normal(0, 3, (T, C))).astype(np.float32)
    mask = rng.random(C) < 0.3
    mask[[3, 5, 70, 71]] = False
    ref[:, mask] = np.nan
    hist[:, mask] = np.nan
    hist[rng.integers(0, T, 40), 3] = np.inf          # several infinite samples: neighbouring order statistics are both inf
    ref[rng.integers(0, T, 40), 70] = -np.inf
    ref[:, 5] = 0.0                                   # 0 / 0 and 0 - 0 next to column 3
    hist[:, 5] = 0.0
    hist[:, 71] = np.nan                              # hist without samples, ref with
    af, hq = (a.get() for a in K_eqm_train(dev, ref, hist, nq, kind))
    eaf, ehq = osdba.eqm_train(ref, hist, nq, kind)
    np.testing.assert_allclose(hq, ehq, rtol=1e-6, equal_nan=True)
    with np.errstate(all="ignore"):
        np.testing.assert_allclose(af, eaf, rtol=1e-5, atol=1e-5 if kind == "+" else 0, equal_nan=True)
    assert np.isnan(af[:, mask]).all() and np.isnan(hq[:, 71]).all() and not np.isnan(hq[:, 3]).any()


def K_eqm_train(dev, ref, hist, nq, kind):
    from xclim_amd import kernels as K

    q = osdba.equally_spaced_nodes(nq) if hasattr(osdba, "equally_spaced_nodes") else (np.arange(nq) + 0.5) / nq
    return K.eqm_train(dev, dev.to_device(ref), dev.to_device(hist), q, kind)

