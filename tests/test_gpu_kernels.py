"""GPU parity: every HIP entry point (through the C ABI) against the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): integer counts / run lengths bit-exact; float quantiles / means <= 1e-6 relative.
"""

import warnings

import numpy as np
import pytest

from oracle import calendar as ocal
from oracle import generic as ogen
from oracle import quantile as oq
from oracle import run_length as orl
from oracle import sdba as osdba
from oracle import synth as osynth
from oracle.timeutil import OTime
from xclim_amd import kernels as K
from xclim_amd.timeaxis import TimeAxis

pytestmark = pytest.mark.gpu

RTOL = 1e-6  # float tolerance stated by the north star


def _field(rng, T, C, nan_frac=0.0, kind="temp"):
    t = np.arange(T)[:, None]
    if kind == "temp":
        x = 288 + 12 * np.sin(2 * np.pi * (t - 100) / 365) + rng.normal(0, 3, (T, C))
    else:
        x = np.where(rng.random((T, C)) < 0.3, rng.gamma(0.8, 8.0, (T, C)) / 86400.0, 0.0)
    x = x.astype(np.float32)
    if nan_frac:
        x[rng.random((T, C)) < nan_frac] = np.nan
    return x


def _times(start, T, calendar="standard"):
    if calendar == "standard":
        return TimeAxis.daily(start, T), OTime.standard(start, T)
    return TimeAxis.daily(start, T, calendar), OTime.noleap(int(start[:4]), T, calendar)


@pytest.mark.parametrize("C", [1, 7, 256, 1000])
@pytest.mark.parametrize("op", [">", "<", ">=", "<=", "==", "!="])
def test_threshold_count_scalar(dev, rng, C, op):
    T = 800
    x = _field(rng, T, C, nan_frac=0.01)
    x[5] = np.float32(290.0)  # exact ties for ==, >=
    ta, ot = _times("2000-03-15", T)
    for freq in ("YS", "MS", "QS-DEC", "YS-JUL"):
        seg, _ = ta.segments(freq)
        cnt, val = K.threshold_count(dev, dev.to_device(x), op, seg, scalar=290.0)
        exp = ogen.count_occurrences(x, 290.0, op, ot, freq)
        np.testing.assert_array_equal(cnt.get(), exp)
        expv = ogen.select_resample_op(x, "count", ot, freq)
        np.testing.assert_array_equal(val.get(), expv)


def test_threshold_count_scalar_promotion(dev, rng):
    """python-float threshold -> fp32 compare; np.float64 threshold -> fp64 compare."""
    T, C = 400, 64
    thr = 1.0 / 86400.0
    x = np.full((T, C), np.float32(thr), dtype=np.float32)  # equals the fp32-rounded threshold
    x[::2] = np.nextafter(np.float32(thr), np.float32(0))
    ta, ot = _times("2001-01-01", T)
    seg, _ = ta.segments("YS")
    d = dev.to_device(x)
    c32, _ = K.threshold_count(dev, d, "<", seg, scalar=thr)
    c64, _ = K.threshold_count(dev, d, "<", seg, scalar=thr, scalar_f64=True)
    np.testing.assert_array_equal(c32.get(), ogen.threshold_count(x, "<", thr, ot, "YS"))
    np.testing.assert_array_equal(c64.get(), ogen.threshold_count(x, "<", np.float64(thr), ot, "YS"))
    assert not np.array_equal(c32.get(), c64.get())


@pytest.mark.parametrize("C", [3, 512])
def test_threshold_count_doy_and_full(dev, rng, C):
    T = 365 * 3 + 1
    x = _field(rng, T, C, nan_frac=0.005)
    ta, ot = _times("2001-01-01", T)
    seg, _ = ta.segments("YS")
    table = (288 + 12 * np.sin(2 * np.pi * (np.arange(366)[:, None] - 100) / 365) + rng.normal(0, 1, (366, C)))
    tidx = (ta.doy - 1).astype(np.int32)
    full = table[tidx]
    d = dev.to_device(x)
    c1, _ = K.threshold_count(dev, d, ">", seg, doy_table=dev.to_device(table), tidx=tidx)
    c2, _ = K.threshold_count(dev, d, ">", seg, full=dev.to_device(full))
    exp = ogen.threshold_count(x, ">", full, ot, "YS")
    np.testing.assert_array_equal(c1.get(), exp)
    np.testing.assert_array_equal(c2.get(), exp)
    t32 = table.astype(np.float32)
    c3, _ = K.threshold_count(dev, d, ">=", seg, doy_table=dev.to_device(t32), tidx=tidx)
    np.testing.assert_array_equal(c3.get(), ogen.threshold_count(x, ">=", t32[tidx], ot, "YS"))


@pytest.mark.parametrize("op", [">", "<", ">=", "<="])
@pytest.mark.parametrize("C,freq", [(200, "YS"), (64, "MS"), (1031, "QS-DEC")])
def test_threshold_count_doy_multi_year_tile_kernel(dev, rng, op, C, freq):
    """tcount.hip (xh_threshold_count_doy on >= 3 years: LDS-resident fp32 thresholds, packed counters): counts and valid
    counts bit-identical to the float64 compare of the reference (gen:301-361), on a leap calendar, ragged column tiles,
    NaN samples, NaN thresholds, thresholds that are exactly a sample / between two floats, a step whose doy the table
    does not hold, and periods that do not cover the whole series."""
    T = 365 * 9 + 2 + 117  # 9 years and a bit: full batches of 256 rows plus a tail
    x = _field(rng, T, C, nan_frac=0.004)
    ta, ot = _times("2001-03-01", T)
    seg, _ = ta.segments(freq)
    seg = np.asarray(seg)[1:-1] if freq == "MS" else np.asarray(seg)  # MS: the first and last month belong to no period
    table = (288 + 12 * np.sin(2 * np.pi * (np.arange(366)[:, None] - 100) / 365) + rng.normal(0, 1, (366, C)))
    tidx = (ta.doy - 1).astype(np.int32)
    tt = rng.integers(0, T, 400)
    cc = rng.integers(0, C, 400)
    table[tidx[tt], cc] = x[tt, cc].astype(np.float64)                     # ties: the threshold IS a sample
    tt2 = rng.integers(0, T, 400)
    cc2 = rng.integers(0, C, 400)
    table[tidx[tt2], cc2] = x[tt2, cc2].astype(np.float64) * (1 + rng.choice([-1e-12, 1e-12], 400))  # just beside one
    table[rng.integers(0, 366, 30), rng.integers(0, C, 30)] = np.nan
    tidx_bad = tidx.copy()
    tidx_bad[[5, 1000, T - 1]] = [-1, 366, 400]
    full = np.where(((tidx_bad >= 0) & (tidx_bad < 366))[:, None], table[np.clip(tidx_bad, 0, 365)], np.nan)
    import operator
    fn = {">": operator.gt, "<": operator.lt, ">=": operator.ge, "<=": operator.le}[op]
    with np.errstate(invalid="ignore"):
        hit = fn(x.astype(np.float64), full)
    ok = ~np.isnan(x)
    P = len(seg) - 1
    exp_c = np.stack([hit[seg[p]:seg[p + 1]].sum(axis=0) for p in range(P)]).astype(np.int32)
    exp_v = np.stack([ok[seg[p]:seg[p + 1]].sum(axis=0) for p in range(P)]).astype(np.int32)
    d = dev.to_device(x)
    trace = dev.start_trace()
    cnt, val = K.threshold_count(dev, d, op, seg, doy_table=dev.to_device(table), tidx=tidx_bad)
    dev.stop_trace()
    assert [n for n, _ in trace if n.startswith("xh_threshold_count")] == ["xh_threshold_count_doy"]
    np.testing.assert_array_equal(cnt.get(), exp_c)
    np.testing.assert_array_equal(val.get(), exp_v)


def test_threshold_count_doy_tile_kernel_empty_periods_and_narrow_counters(dev, rng):
    """tcount.hip: periods without a step (equal consecutive offsets) count 0 / 0; 8-bit counter halves (every period
    shorter than 256 steps) and 16-bit ones (a period of 400 steps) give the same counts as the per-period kernel."""
    T, C = 4000, 130
    x = _field(rng, T, C, nan_frac=0.01)
    table = 288 + rng.normal(0, 4, (366, C))
    tidx = (np.arange(T) % 366).astype(np.int32)
    for seg in (np.array([1455, 1488], dtype=np.int64),   # ONE short period inside the 256 rows a batch spans, no period around it
                np.array([0, 200, 200, 200, 431, 600, 855, 1000, 1000, 1255, T - 7], dtype=np.int64),     # narrow, gaps at both ends
                np.array([3, 403, 500, 900, 1300, 1300, 1700, 2100, 2500, 2900, 3300, 3700, T], dtype=np.int64)):  # a 400-step period
        hit = x.astype(np.float64) <= table[tidx]
        ok = ~np.isnan(x)
        P = len(seg) - 1
        exp_c = np.stack([hit[seg[p]:seg[p + 1]].sum(axis=0) for p in range(P)]).astype(np.int32)
        exp_v = np.stack([ok[seg[p]:seg[p + 1]].sum(axis=0) for p in range(P)]).astype(np.int32)
        trace = dev.start_trace()
        cnt, val = K.threshold_count(dev, dev.to_device(x), "<=", seg, doy_table=dev.to_device(table), tidx=tidx)
        dev.stop_trace()
        assert [n for n, _ in trace if n.startswith("xh_threshold_count")] == ["xh_threshold_count_doy"]
        np.testing.assert_array_equal(cnt.get(), exp_c)
        np.testing.assert_array_equal(val.get(), exp_v)


def test_domain_count(dev, rng):
    T, C = 730, 300
    x = _field(rng, T, C, nan_frac=0.01)
    ta, ot = _times("2000-01-01", T)
    seg, _ = ta.segments("MS")
    c, _ = K.domain_count(dev, dev.to_device(x), ">", 285.0, "<=", 295.0, "and", seg)
    np.testing.assert_array_equal(c.get(), ogen.domain_count(x, np.float32(285.0), np.float32(295.0), ot, "MS"))


@pytest.mark.parametrize("reducer", ["sum", "mean", "min", "max", "std", "var", "count", "argmin", "argmax"])
@pytest.mark.parametrize("C", [5, 1024])
def test_resample_reduce(dev, rng, reducer, C):
    T = 900
    x = _field(rng, T, C, nan_frac=0.02)
    x[:40, 0] = np.nan  # an all-NaN month
    ta, ot = _times("1999-11-20", T)
    for freq in ("YS", "MS"):
        seg, _ = ta.segments(freq)
        out, _ = K.resample_reduce(dev, dev.to_device(x), reducer, seg)
        exp = ogen.select_resample_op(x, reducer, ot, freq)
        if reducer in ("count", "argmin", "argmax"):
            np.testing.assert_array_equal(out.get(), exp)
        else:
            np.testing.assert_allclose(out.get(), exp, rtol=RTOL, atol=1e-30 if reducer not in ("std", "var") else 1e-6,
                                       equal_nan=True)


@pytest.mark.parametrize("window,center", [(5, True), (3, False), (14, True), (4, True)])
@pytest.mark.parametrize("reducer", ["sum", "mean", "min", "max", "std"])
def test_rolling_reduce(dev, rng, window, center, reducer):
    T, C = 120, 70
    x = _field(rng, T, C, nan_frac=0.02)
    out = K.rolling_reduce(dev, dev.to_device(x), window, reducer, center)
    exp = ogen.rolling(x, window, reducer, center)
    np.testing.assert_allclose(out.get(), exp, rtol=RTOL, atol=2e-6 if reducer == "std" else 0, equal_nan=True)


def _mask(rng, T, C, p=0.6, nan_frac=0.0):
    m = (rng.random((T, C)) < p).astype(np.float32)
    # long runs
    m[10:60, : C // 3] = 1
    if nan_frac:
        m[rng.random((T, C)) < nan_frac] = np.nan
    return m


@pytest.mark.parametrize("index", ["first", "last"])
@pytest.mark.parametrize("nan_frac", [0.0, 0.05])
def test_cumsum_reset_and_rle(dev, rng, index, nan_frac):
    T, C = 200, 130
    m = _mask(rng, T, C, nan_frac=nan_frac)
    m[:, 0] = np.nan  # all-NaN column -> 0 (tests/test_run_length.py:89-91)
    m[:, 1] = 1
    m[:, 2] = 0
    d = dev.to_device(m)
    np.testing.assert_array_equal(K.cumsum_reset(dev, d, index).get(), orl.cumsum_reset(m, index))
    np.testing.assert_array_equal(K.rle(dev, d, index).get(), orl.rle(m, index))


@pytest.mark.parametrize("stat", ["max", "min", "sum", "count", "mean", "std"])
@pytest.mark.parametrize("index", ["first", "last"])
@pytest.mark.parametrize("cut", [True, False])
def test_run_stats_mask(dev, rng, stat, index, cut):
    T, C = 800, 260
    ta, ot = _times("2000-02-10", T)
    for nan_frac, window, freq in ((0.0, 1, "YS"), (0.03, 3, "MS"), (0.03, 1, "QS-DEC")):
        m = _mask(rng, T, C, nan_frac=nan_frac)
        seg, _ = ta.segments(freq)
        # the N-D path of the reference (rle quirk: a run next to a NaN step loses its length) ...
        out, _ = K.run_stats(dev, dev.to_device(m), stat, window, seg, cut=cut, index=index)
        exp = orl.resample_and_rl(m, cut, orl.rle_statistics, time=ot, freq=freq, reducer=stat, window=window, index=index,
                                  ufunc_1dim=False)
        if stat in ("mean", "std"):
            np.testing.assert_allclose(out.get(), exp, rtol=RTOL, atol=1e-5 if stat == "std" else 0)
        else:
            np.testing.assert_array_equal(out.get(), exp)
        # ... and the 1-D ufunc path it takes for grids under 9000 cells when the runs are cut first (index "first")
        if cut and index == "first":
            out, _ = K.run_stats(dev, dev.to_device(m), stat, window, seg, cut=True, index=index, one_dim="stat")
            exp = orl.resample_and_rl(m, True, orl.rle_statistics, time=ot, freq=freq, reducer=stat, window=window,
                                      ufunc_1dim=True)
            np.testing.assert_allclose(out.get(), exp, rtol=RTOL, atol=1e-5 if stat == "std" else 0, equal_nan=True)


@pytest.mark.parametrize("window", [1, 2, 5])
@pytest.mark.parametrize("index", ["first", "last"])
def test_windowed_run_count_events(dev, rng, window, index):
    T, C = 500, 100
    ta, ot = _times("2003-01-01", T)
    m = _mask(rng, T, C, nan_frac=0.02)
    seg, _ = ta.segments("MS")
    d = dev.to_device(m)
    for cut in (True, False):
        ev, _ = K.run_stats(dev, d, "count", window, seg, cut=cut, index=index)
        exp_ev = orl.resample_and_rl(m, cut, orl.windowed_run_events, window, time=ot, freq="MS", index=index, ufunc_1dim=False)
        np.testing.assert_array_equal(ev.get(), exp_ev)
        stat = "plainsum" if (window == 1 and cut) else "sum"
        cn, _ = K.run_stats(dev, d, stat, window, seg, cut=cut, index=index)
        exp_cn = orl.resample_and_rl(m, cut, orl.windowed_run_count, window, time=ot, freq="MS", index=index, ufunc_1dim=False)
        np.testing.assert_array_equal(cn.get(), exp_cn)
        if cut and index == "first":  # the 1-D ufunc path (grids under 9000 cells): runs next to a NaN keep their length
            ev, _ = K.run_stats(dev, d, "count", window, seg, cut=True, one_dim=True)
            np.testing.assert_array_equal(ev.get(), orl.resample_and_rl(m, True, orl.windowed_run_events, window, time=ot,
                                                                        freq="MS", ufunc_1dim=True))
            cn, _ = K.run_stats(dev, d, "sum", window, seg, cut=True, one_dim=True)
            np.testing.assert_array_equal(cn.get(), orl.resample_and_rl(m, True, orl.windowed_run_count, window, time=ot,
                                                                        freq="MS", ufunc_1dim=True))


@pytest.mark.parametrize("window", [1, 3, 7])
@pytest.mark.parametrize("cut", [True, False])
def test_first_last_run(dev, rng, window, cut):
    T, C = 730, 90
    ta, ot = _times("2001-01-01", T)
    m = _mask(rng, T, C, p=0.5, nan_frac=0.02)
    m[:, 0] = 1  # all-True quirk (rl:603-605)
    m[:, 1] = 0
    m[300:420, 2] = 1  # a run covering a whole month
    seg, _ = ta.segments("MS")
    d = dev.to_device(m)
    if window == 1 and not cut:
        pytest.skip("window == 1 with freq always maps per group in the reference (rl:618-621)")
    f, _ = K.run_stats(dev, d, "first", window, seg, cut=cut)
    l, _ = K.run_stats(dev, d, "last", window, seg, cut=cut)
    exp_f = orl.resample_and_rl(m, cut, orl.first_run, window, time=ot, freq="MS")
    exp_l = orl.resample_and_rl(m, cut, orl.last_run, window, time=ot, freq="MS")
    np.testing.assert_array_equal(f.get(), exp_f)
    np.testing.assert_array_equal(l.get(), exp_l)


def test_cdd_fused(dev, rng):
    """maximum_consecutive_dry_days: fused compare + rle max, before/after resampling (26 vs 30 style)."""
    T, C = 1461, 500
    pr = _field(rng, T, C, nan_frac=0.001, kind="pr")
    thr = 1.0 / 86400.0
    ta, ot = _times("2000-01-01", T)
    seg, _ = ta.segments("YS")
    d = dev.to_device(pr)
    for cut in (True, False):
        out, val = K.run_stats(dev, d, "max", 1, seg, cut=cut, fused_op="<", thresh=thr)
        exp = ogen.spell_length_statistics(pr, thr, 1, None, "<", "max", ot, "YS", resample_before_rl=cut)
        np.testing.assert_array_equal(out.get(), exp)
        np.testing.assert_array_equal(val.get(), ogen.select_resample_op(pr, "count", ot, "YS"))


@pytest.mark.parametrize("N,C", [(1, 10), (2, 10), (5, 300), (8, 64), (13, 70), (30, 129), (150, 200), (365, 40), (600, 70), (1000, 40),
                                 (2500, 20)])
@pytest.mark.parametrize("ab", [(1.0, 1.0), (1 / 3, 1 / 3)])
def test_nan_quantile(dev, rng, N, C, ab):
    x = rng.normal(0, 1, (N, C)).astype(np.float32)
    x[rng.random((N, C)) < 0.1] = np.nan
    if C > 3:
        x[:, 0] = np.nan
        x[1:, 1] = np.nan
        x[:, 2] = 1.5
    q = np.array([0.0, 0.1, 0.5, 0.9, 0.99, 1.0])
    out = K.nan_quantile(dev, dev.to_device(x), q, *ab).get()
    exp = oq.nan_quantile(x, q, 0, *ab)
    np.testing.assert_allclose(out, exp, rtol=1e-12, atol=0, equal_nan=True)
    # sample-minor layout (what apply_ufunc hands calc_perc)
    out2 = K.nan_quantile(dev, dev.to_device(np.ascontiguousarray(x.T)), q, *ab, sample_axis=1).get()
    np.testing.assert_allclose(out2, exp, rtol=1e-12, atol=0, equal_nan=True)


@pytest.mark.parametrize("nyears,window,C,calendar", [(1, 5, 260, "noleap"), (1, 5, 64, "standard"), (3, 5, 33, "standard"),
                                                     (6, 5, 70, "noleap"), (9, 7, 20, "standard"), (30, 5, 24, "noleap"),
                                                     (2, 4, 16, "noleap"), (20, 31, 40, "noleap"), (30, 31, 20, "standard")])
def test_percentile_doy(dev, rng, nyears, window, C, calendar):
    start = "2000-01-01"
    T = 365 * nyears + (nyears + 3) // 4 if calendar == "standard" else 365 * nyears
    x = _field(rng, T, C, nan_frac=0.01)
    ta, ot = _times(start, T, calendar)
    tb, years, doys = ta.doy_table()
    per = [10.0, 50.0, 90.0]
    out = K.percentile_doy(dev, dev.to_device(x), tb, window, per).get()  # (nper, ndoy, C)
    # oracle without the 366 adjustment: call the stacked calc_perc directly
    rr = ocal.rolling_construct_center(x, window)
    stack = np.full((len(doys), len(years), C, window), np.nan, dtype=np.float32)
    stack[np.searchsorted(doys, ot.doy), np.searchsorted(years, ot.year)] = rr
    stack = np.moveaxis(stack, 1, -2).reshape(len(doys), C, len(years) * window)
    exp = oq.calc_perc(stack, per, 1 / 3, 1 / 3)  # (ndoy, C, nper)
    np.testing.assert_allclose(out, np.moveaxis(exp, -1, 0), rtol=1e-12, atol=0, equal_nan=True)


@pytest.mark.parametrize("nyears,window,calendar,nan_frac", [(30, 5, "standard", 0.0), (40, 3, "noleap", 0.3), (12, 7, "standard", 0.05),
                                                           (64, 5, "noleap", 0.01), (8, 5, "standard", 0.9)])
def test_percentile_doy_merge_path(dev, rng, nyears, window, calendar, nan_frac):
    """Multi-year base periods: per-day sorted lists + W-way tail merge (k_pdoy_merge) + LDS fallback on the
    irregular days of year, all percentiles from both ends."""
    C = 70
    T = 365 * nyears + (nyears + 3) // 4 if calendar == "standard" else 365 * nyears
    x = _field(rng, T, C, nan_frac=nan_frac)
    x[:, 0] = np.nan
    x[:, 1] = 280.0
    ta, ot = _times("2000-01-01", T, calendar)
    tb, years, doys = ta.doy_table()
    per = [0.0, 2.0, 10.0, 50.0, 75.0, 90.0, 99.0, 100.0]
    out = K.percentile_doy(dev, dev.to_device(x), tb, window, per).get()
    rr = ocal.rolling_construct_center(x, window)
    stack = np.full((len(doys), len(years), C, window), np.nan, dtype=np.float32)
    stack[np.searchsorted(doys, ot.doy), np.searchsorted(years, ot.year)] = rr
    stack = np.moveaxis(stack, 1, -2).reshape(len(doys), C, len(years) * window)
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exp = oq.calc_perc(stack, per, 1 / 3, 1 / 3)
    np.testing.assert_allclose(out, np.moveaxis(exp, -1, 0), rtol=1e-12, atol=0, equal_nan=True)


@pytest.mark.parametrize("nyears,calendar,per,nan_frac", [(30, "noleap", 90.0, 0.0), (30, "noleap", 10.0, 0.0), (30, "standard", 90.0, 0.0),
                                                         (7, "noleap", 90.0, 0.0), (7, "noleap", 5.0, 0.0), (33, "noleap", 95.0, 0.0),
                                                         (30, "noleap", 90.0, 0.02), (31, "standard", 10.0, 0.3), (12, "noleap", 99.0, 0.0)])
def test_percentile_doy_quad_kernel(dev, rng, monkeypatch, nyears, calendar, per, nan_frac):
    """One percentile on a window of 5 over a multi-year base period = k_pdoy_quad's fast path (pdoy_quad.hip: quad sharing,
    NaN-free day-sets skip the conversion, padding / absent days read from constant rows): against the oracle, and
    bit-identical to k_pdoy_top16 (XH_PDOY_QUAD=0), on a column count that is no multiple of 64, with the special values."""
    C, window = 150, 5
    T = 365 * nyears + (nyears + 3) // 4 if calendar == "standard" else 365 * nyears
    x = _field(rng, T, C, nan_frac=nan_frac)
    x[:, 0] = np.nan           # a cell without data
    x[:, 1] = 280.0            # ties everywhere
    x[::7, 2] = np.inf         # infinities of both signs in one day-set window
    x[3::11, 2] = -np.inf
    x[: 365 * 2, 3] = np.nan   # two years missing: another valid count than the neighbours
    x[100:160, 4] = np.nan     # a gap in one year
    x[:, 5] = -x[:, 6]
    ta, ot = _times("2000-01-01", T, calendar)
    tb, years, doys = ta.doy_table()
    d_x = dev.to_device(x)
    out = K.percentile_doy(dev, d_x, tb, window, [per]).get()
    rr = ocal.rolling_construct_center(x, window)  # (the kernel's table has no 366 -> 365 adjustment: stacked calc_perc)
    stack = np.full((len(doys), len(years), C, window), np.nan, dtype=np.float32)
    stack[np.searchsorted(doys, ot.doy), np.searchsorted(years, ot.year)] = rr
    stack = np.moveaxis(stack, 1, -2).reshape(len(doys), C, len(years) * window)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exp = oq.calc_perc(stack, [per], 1 / 3, 1 / 3)
    np.testing.assert_allclose(out, np.moveaxis(exp, -1, 0), rtol=1e-12, atol=0, equal_nan=True)
    monkeypatch.setenv("XH_DIAGNOSTICS", "1")
    monkeypatch.setenv("XH_PDOY_QUAD", "0")
    ref = K.percentile_doy(dev, d_x, tb, window, [per]).get()
    np.testing.assert_array_equal(out, ref)
    monkeypatch.setenv("XH_PDOY_QUAD", "1")
    monkeypatch.setenv("XH_PDOY_CHUNK", "10")  # other chunk boundaries, same numbers
    np.testing.assert_array_equal(K.percentile_doy(dev, d_x, tb, window, [per]).get(), out)


@pytest.mark.parametrize("nyears,window,calendar,nan_frac", [(30, 5, "noleap", 0.0), (30, 5, "standard", 0.02), (12, 7, "noleap", 0.3),
                                                           (20, 3, "standard", 0.0), (9, 5, "noleap", 0.9)])
def test_percentile_doy_walk_kernel(dev, rng, monkeypatch, nyears, window, calendar, nan_frac):
    """Central percentiles on a multi-year base period = k_pdoy_walk (pdoy_walk.hip: sorted day-set lists in LDS, a split
    per percentile that walks from day to day): against the oracle and bit-identical to the pop-from-one-end kernel it
    replaces (k_pdoy_merge, XH_PDOY_WALK=0); more percentiles than one launch holds; the special values."""
    C = 130
    T = 365 * nyears + (nyears + 3) // 4 if calendar == "standard" else 365 * nyears
    x = _field(rng, T, C, nan_frac=nan_frac)
    x[:, 0] = np.nan
    x[:, 1] = 280.0
    x[::7, 2] = np.inf
    x[3::11, 2] = -np.inf
    x[: 365 * 2, 3] = np.nan
    x[:, 4] = np.round(x[:, 4])  # ties
    ta, ot = _times("2000-01-01", T, calendar)
    tb, years, doys = ta.doy_table()
    per = [20.0, 33.0, 50.0, 66.6, 75.0, 40.0]
    d_x = dev.to_device(x)
    out = K.percentile_doy(dev, d_x, tb, window, per).get()
    rr = ocal.rolling_construct_center(x, window)
    stack = np.full((len(doys), len(years), C, window), np.nan, dtype=np.float32)
    stack[np.searchsorted(doys, ot.doy), np.searchsorted(years, ot.year)] = rr
    stack = np.moveaxis(stack, 1, -2).reshape(len(doys), C, len(years) * window)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exp = oq.calc_perc(stack, per, 1 / 3, 1 / 3)
    np.testing.assert_allclose(out, np.moveaxis(exp, -1, 0), rtol=1e-12, atol=0, equal_nan=True)
    monkeypatch.setenv("XH_DIAGNOSTICS", "1")
    monkeypatch.setenv("XH_PDOY_WALK", "0")
    np.testing.assert_array_equal(K.percentile_doy(dev, d_x, tb, window, per).get(), out)
    monkeypatch.setenv("XH_PDOY_WALK", "1")
    monkeypatch.setenv("XH_PDOY_CHUNK", "11")
    np.testing.assert_array_equal(K.percentile_doy(dev, d_x, tb, window, per).get(), out)


def test_infinities_follow_the_nanmax_rule(dev, rng):
    """utl:552-554: a NaN interpolation (inf - inf between two order statistics) becomes the slice's nanmax — in the one-shot
    quantile (also where the virtual index is below 0 and both neighbours are slot 0), in the series quantiles of every
    length class, and in the one-year percentile_doy kernel's minimum short-cut (found by tools/fuzz_inf.py)."""
    def field(T, C):
        x = np.round(rng.normal(10, 4, (T, C)), 1).astype(np.float32)
        r = rng.random((T, C))
        x[r < 0.02] = np.inf
        x[(r >= 0.02) & (r < 0.04)] = -np.inf
        x[:, 0] = np.inf
        x[:, 1] = -np.inf
        x[::2, 2], x[1::2, 2] = np.inf, -np.inf
        x[: T // 2, 3] = -np.inf
        x[:, 4] = np.nan
        return x

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for n in (3, 17, 150):
            x = field(n, 40)
            q = np.array([0.01, 0.3, 0.5, 0.99])
            for ab in ((1.0, 1.0), (1 / 3, 1 / 3)):
                np.testing.assert_allclose(K.nan_quantile(dev, dev.to_device(x), q, *ab).get(), oq.nan_quantile(x, q, 0, *ab),
                                           rtol=1e-12, equal_nan=True, err_msg=f"nan_quantile n={n} {ab}")
        for T in (365, 800, 3650, 5000, 33000):
            x = field(T, 40)
            q = osdba.equally_spaced_nodes(20)
            exp = oq.nan_quantile(x, q, axis=0, alpha=1.0, beta=1.0).astype(np.float32)
            np.testing.assert_array_equal(K.quantile_series(dev, dev.to_device(x), q).get(), exp, err_msg=f"quantile_series T={T}")
        x = field(365, 40)
        ta, ot = _times("2001-01-01", 365, "noleap")
        tb, years, doys = ta.doy_table()
        for per in ([10.0], [90.0], [10.0, 50.0, 90.0]):
            got = K.percentile_doy(dev, dev.to_device(x), tb, 5, per).get()
            rr = ocal.rolling_construct_center(x, 5)
            stack = np.full((len(doys), 1, 40, 5), np.nan, dtype=np.float32)
            stack[np.searchsorted(doys, ot.doy), 0] = rr
            stack = np.moveaxis(stack, 1, -2).reshape(len(doys), 40, 5)
            exp = np.moveaxis(oq.calc_perc(stack, per, 1 / 3, 1 / 3), -1, 0)
            np.testing.assert_allclose(got, exp, rtol=1e-12, equal_nan=True, err_msg=f"percentile_doy one year {per}")


@pytest.mark.parametrize("nyears", [4, 12])
def test_percentile_doy_virtual_time_map(dev, rng, nyears):
    """vmap: percentile_doy of a series in which one year was replaced by another, without copying the data."""
    C, window = 40, 5
    T = 365 * nyears
    x = _field(rng, T, C, nan_frac=0.01)
    ta, ot = _times("2001-01-01", T, "noleap")
    tb, years, doys = ta.doy_table()
    vmap = np.arange(T, dtype=np.int32)
    vmap[365 * 1 : 365 * 2] = np.arange(365 * 3, 365 * 4)  # year 1 <- year 3
    vmap[365 * 2 + 59] = -1  # and a day the replica lacks
    out = K.percentile_doy(dev, dev.to_device(x), tb, window, [10.0, 90.0], vmap=vmap).get()
    xm = np.where(vmap[:, None] >= 0, x[np.clip(vmap, 0, None)], np.nan).astype(np.float32)
    exp, _ = ocal.percentile_doy(xm, ot, window, [10.0, 90.0])
    np.testing.assert_allclose(out, np.moveaxis(exp, -1, 0), rtol=1e-12, atol=0, equal_nan=True)


def test_doy_interp(dev, rng):
    C = 50
    src = rng.normal(280, 5, (365, C))
    src[100:104, 3] = np.nan
    src[:2, 4] = np.nan
    src[-3:, 5] = np.nan
    exp, target = ocal.interpolate_doy_calendar(src, np.arange(1, 366), 366, 1)
    from xclim_amd.calendar import doy_interp_tables

    i0, i1, dxn, dxs = doy_interp_tables(365, 366, 1)
    out = K.doy_interp(dev, dev.to_device(src, dtype=np.float64), i0, i1, dxn, dxs).get()
    np.testing.assert_allclose(out, exp, rtol=1e-13, atol=0, equal_nan=True)
    # a doy that never occurs in the series (365 days from Feb 28 of a leap year: no doy 58): interpolate_na works in
    # the dayofyear COORDINATE, so the NaN gap next to the missing doy is filled with non-uniform weights
    doys = np.array([d for d in range(1, 366) if d != 58], dtype=np.float64)
    src2 = rng.normal(280, 5, (364, C))
    src2[54:58, 7] = np.nan  # doys 55, 56, 57, 59
    src2[56, 8] = np.nan
    exp2, _ = ocal.interpolate_doy_calendar(src2, doys, 366, 1)
    i0, i1, dxn, dxs = doy_interp_tables(364, 366, 1)
    out2 = K.doy_interp(dev, dev.to_device(src2, dtype=np.float64), i0, i1, dxn, dxs, xsrc=doys).get()
    np.testing.assert_allclose(out2, exp2, rtol=1e-13, atol=0, equal_nan=True)


@pytest.mark.parametrize("T,C", [(365, 100), (40, 7), (1000, 33), (5000, 6), (10950, 5)])
def test_quantile_series(dev, rng, T, C):
    x = _field(rng, T, C, nan_frac=0.01)
    x[:, 0] = np.nan
    q = osdba.equally_spaced_nodes(20)
    exp = osdba.quantile(x, q)
    out = K.quantile_series(dev, dev.to_device(x), q).get()
    np.testing.assert_allclose(out, exp, rtol=RTOL, atol=0, equal_nan=True)
    out2 = K.quantile_series(dev, dev.to_device(np.ascontiguousarray(x.T)), q, time_axis=1).get()
    np.testing.assert_allclose(out2, exp, rtol=RTOL, atol=0, equal_nan=True)


@pytest.mark.parametrize("T", [40, 365, 512, 600, 1000, 1200, 3000, 5000, 10950])
@pytest.mark.parametrize("kind", ["pr", "pr_skewed", "two_values", "negative_floor", "clustered", "constant", "nan_heavy"])
def test_quantile_series_hard_distributions(dev, rng, T, kind):
    """Every selection kernel (T <= 512 grouped, <= 1024 one workgroup per column, longer: list-free) on the inputs that
    stress the key histogram: exact-zero floors (the smallest key owns bin 0), skewed wet-day amounts and big
    non-constant bins (in-bin quickselect / bisection fallback), two-valued and constant columns, a tied NEGATIVE minimum,
    mostly-NaN columns."""
    C = 9
    if kind == "pr":
        x = _field(rng, T, C, kind="pr")
    elif kind == "pr_skewed":
        u = rng.random((T, C))
        x = np.where(rng.random((T, C)) < 0.6, (u * u * u) * (40.0 / 86400.0), 0.0).astype(np.float32)
        x[:, 3] = np.where(rng.random(T) < 0.02, 1.0, 0.0)  # a handful of wet days
    elif kind == "two_values":
        x = np.where(rng.random((T, C)) < 0.5, 2.5, -1.0).astype(np.float32)
        x[:, 2] = np.where(np.arange(T) == T // 2, 7.0, 7.5)  # one copy of the smaller value
    elif kind == "negative_floor":
        x = np.maximum(rng.normal(0, 1, (T, C)), -0.25).astype(np.float32)  # censored: ~40 % of the days at -0.25
        x[rng.random((T, C)) < 0.01] = np.nan
    elif kind == "clustered":
        x = (1.0 + rng.integers(0, 50, (T, C)) * 1.1920929e-07).astype(np.float32)  # 50 adjacent float32 values
        x[rng.random((T, C)) < 0.02] = 1.0e6  # outliers stretch the key range: the cluster lands in one bin
    elif kind == "constant":
        x = np.full((T, C), 3.25, np.float32)
        x[:, 1] = np.linspace(0, 1, T, dtype=np.float32)
    else:
        x = _field(rng, T, C)
        x[rng.random((T, C)) < 0.97] = np.nan
        x[:, 0] = np.nan
    q = np.concatenate([osdba.equally_spaced_nodes(20), [0.0, 1.0, 0.5]])
    q.sort()
    exp = osdba.quantile(x, q)
    out = K.quantile_series(dev, dev.to_device(np.ascontiguousarray(x.T)), q, time_axis=1).get()
    np.testing.assert_allclose(out, exp, rtol=RTOL, atol=0, equal_nan=True)
    out2 = K.quantile_series(dev, dev.to_device(x), q).get()
    np.testing.assert_allclose(out2, exp, rtol=RTOL, atol=0, equal_nan=True)


@pytest.mark.parametrize("per,op", [(90.0, ">"), (95.0, ">="), (10.0, "<"), (5.0, "<=")])
@pytest.mark.parametrize("freq", ["YS", "MS", "QS-DEC"])
def test_percentile_doy_count_multi_year(dev, rng, per, op, freq):
    """xh_percentile_doy_count on a 30-year base period (k_pdoy_top16<.., COUNT>): counts and valid counts identical to
    percentile_doy -> threshold_count (fp64 compare) and to the oracle; the fp32 threshold trick (largest float <= r /
    smallest float >= r) must agree with the fp64 compare on ties, which rounded values provoke."""
    T, C = 365 * 30, 150
    ta, ot = _times("1981-01-01", T, "noleap")
    x = _field(rng, T, C, nan_frac=0.002)
    x[:, 5:20] = np.round(x[:, 5:20] * 2) / 2   # many samples equal to the percentile
    x[:, 3] = np.nan
    x[: T // 2, 4] = np.nan
    tb, years, doys = ta.doy_table()
    seg, _ = ta.segments(freq)
    P = len(seg) - 1
    tidx = np.searchsorted(doys, ta.doy).astype(np.int32)
    xd = dev.to_device(x)
    p = K.percentile_doy(dev, xd, tb, 5, [per])
    cnt, val = K.threshold_count(dev, xd, op, seg, doy_table=p.reshape(len(doys), C), tidx=tidx)
    period = (np.searchsorted(seg, tb, side="right") - 1).astype(np.int32)
    period[tb < 0] = -1
    fused = K.percentile_doy_count(dev, xd, tb, 5, per, op, period, P)
    assert fused is not None
    np.testing.assert_array_equal(fused[0].get(), cnt.get())
    np.testing.assert_array_equal(fused[1].get(), val.get())
    pe, d2 = ocal.percentile_doy(x, ot, 5, per)
    thresh = ocal.resample_doy(pe[..., 0], d2, ot)
    np.testing.assert_array_equal(fused[0].get(), ogen.threshold_count(x, op, thresh, ot, freq))
    # a central percentile is not covered by the register top-16 kernel: the caller falls back to the chain
    assert K.percentile_doy_count(dev, xd, tb, 5, 50.0, ">", period, P) is None


@pytest.mark.parametrize("T", [360, 361, 364, 365, 366])
@pytest.mark.parametrize("nq", [1, 20, 50, 64])
def test_quantile_series_one_year_register_sort(dev, rng, T, nq):
    """k_select_regsort (select3.hip): one-year daily series from the time-major view — two lanes per column, keys in
    registers, comparator network.  Clean columns take the walk with the shared rank table, all-NaN columns ride along,
    a tile with ONE partially-NaN column takes the per-lane path; ties, negative NaNs, -0.0, a ragged last
    tile (C % 32 != 0) and the end-point quantiles are covered.  Bitwise against the oracle (same fp64 lerp)."""
    C = 32 * 9 + 5
    x = _field(rng, T, C)
    x[:, 3] = np.nan                                   # all-NaN column inside a clean tile (regular path)
    x[:, 40:44] = np.round(x[:, 40:44])                # heavy ties
    x[:, 45] = -np.abs(x[:, 45]) * 0.0                 # all -0.0
    x[:, 46] = np.where(rng.random(T) < 0.5, -1.5, 2.5)
    x[:, 70] = np.float32(7.25)                        # constant
    x[T // 3, 100] = np.nan                            # one NaN -> its tile is irregular
    x[:, 130] = np.where(rng.random(T) < 0.9, np.nan, x[:, 130])   # mostly NaN
    x[: T - 1, 131] = np.nan                           # a single valid sample
    neg_nan = np.frombuffer(np.uint32(0xFFC00001).tobytes(), np.float32)[0]
    x[7, 200] = neg_nan                                # sign-bit NaN must still sort last
    x[:, C - 2] = np.nan                               # in the ragged tile
    q = np.linspace(0.0, 1.0, nq) if nq > 1 else np.array([0.5])
    if nq == 20:
        q = osdba.equally_spaced_nodes(20)
    exp = osdba.quantile(x, q)
    out = K.quantile_series(dev, dev.to_device(x), q).get()
    np.testing.assert_array_equal(out, exp.astype(np.float32))


def test_quantile_series_register_sort_matches_histogram_kernels(dev, rng, monkeypatch):
    """The same clean field through both time-major short-series kernels (diagnostic switch) and through the NaN path of
    the register kernel: identical bits."""
    T, C = 365, 4099
    x = _field(rng, T, C)
    q = osdba.equally_spaced_nodes(20)
    a = K.quantile_series(dev, dev.to_device(x), q).get()
    monkeypatch.setenv("XH_DIAGNOSTICS", "1")
    monkeypatch.setenv("XH_SELECT_NOREGSORT", "1")
    b = K.quantile_series(dev, dev.to_device(x), q).get()
    monkeypatch.delenv("XH_SELECT_NOREGSORT")
    monkeypatch.setenv("XH_REGSORT_IRREGULAR", "1")
    c = K.quantile_series(dev, dev.to_device(x), q).get()
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(a, c)
    np.testing.assert_array_equal(a, osdba.quantile(x, q).astype(np.float32))


@pytest.mark.parametrize("T", [1025, 1300, 4000, 10950, 20000])
@pytest.mark.parametrize("nq", [1, 20, 32])
def test_quantile_series_two_pass_histogram(dev, rng, T, nq):
    """select4.hip (k_hs_sample -> k_hs_hist -> k_hs_collect): long time-major series in two streaming passes.  Clean,
    NaN-sprinkled, all-NaN, constant, two-valued, precipitation-like (the dry days own the pure "== lo" bin), saturated
    at the maximum (pure "== hi" bin), tied / negative / -0.0 / sign-bit-NaN columns, a single valid sample,
    a ragged last tile (C % 32 != 0).  Bitwise against the oracle (same fp64 lerp) and against the transposed pipeline."""
    C = 32 * 5 + 7
    x = _field(rng, T, C, nan_frac=0.002)
    x[:, 3] = np.nan
    x[:, 4] = np.float32(7.25)
    x[:, 5] = np.where(rng.random(T) < 0.5, -1.5, 2.5)
    x[:, 6:10] = _field(rng, T, 4, kind="pr")
    x[:, 10] = np.minimum(x[:, 10], np.float32(290.0))          # a third of the days at the cap
    x[:, 11] = np.round(x[:, 11] * 4) / 4                          # quarter-degree steps: heavy ties everywhere
    x[:, 12] = -np.abs(x[:, 12]) * 0.0
    x[:, 13] = x[:, 13] - 288                                      # values around zero, both signs
    x[: T - 1, 14] = np.nan
    x[:, 15] = np.where(rng.random(T) < 0.97, np.nan, x[:, 15])
    neg_nan = np.frombuffer(np.uint32(0xFFC00001).tobytes(), np.float32)[0]
    x[7, 16] = neg_nan
    x[:, 18] = np.where(np.arange(T) % 2 == 0, 1.0, 1.0000001)      # two ADJACENT float32 values
    x[:, C - 2] = np.nan
    q = np.linspace(0.0, 1.0, nq) if nq > 1 else np.array([0.5])
    if nq == 20:
        q = osdba.equally_spaced_nodes(20)
    exp = osdba.quantile(x, q).astype(np.float32)
    out = K.quantile_series(dev, dev.to_device(x), q).get()
    np.testing.assert_array_equal(out, exp)


def test_quantile_series_two_pass_matches_transposed_pipeline(dev, rng, monkeypatch):
    """Same field through select4.hip and (diagnostic switch) through the transposed column kernels: identical bits; the
    flagged-column path (clustered values: > 512 candidates) and a non-16-byte-aligned view ride along."""
    T, C = 3000, 2051
    x = _field(rng, T, C, nan_frac=0.001)
    x[:, 100:140] = (1.0 + rng.integers(0, 50, (T, 40)) * 1.1920929e-07).astype(np.float32)
    x[rng.random((T, C)) < 0.0005] = 1.0e6
    q = osdba.equally_spaced_nodes(20)
    exp = osdba.quantile(x, q).astype(np.float32)
    x[5, 17], x[9, 17], x[11, 18] = np.inf, -np.inf, np.inf  # (+-inf: compared between the GPU paths only)
    xd = dev.to_device(x)
    a = K.quantile_series(dev, xd, q).get()
    monkeypatch.setenv("XH_DIAGNOSTICS", "1")
    monkeypatch.setenv("XH_SELECT_NOHIST", "1")
    b = K.quantile_series(dev, xd, q).get()
    np.testing.assert_array_equal(a, b)
    keep = np.ones(C, bool)
    keep[17:19] = False
    np.testing.assert_array_equal(a[:, keep], exp[:, keep])


@pytest.mark.parametrize("T", [40000, 55152])
def test_quantile_series_beyond_32768_steps(dev, rng, T):
    """1950-2100 daily = 55 152 steps: the streaming passes of select4.hip take any T <= 65535 (u16 counters), and what
    they hand back (heavily tied columns) goes to the radix select of select5.hip — round 3 refused T > 32768.  Bitwise
    against the oracle; the time-minor layout (radix select for every column) and nq > 32 (two radix sweeps) ride along."""
    C = 64 * 3 + 5
    x = _field(rng, T, C, nan_frac=0.002)
    x[:, 3] = np.nan
    x[:, 4] = np.float32(7.25)
    x[:, 5] = np.where(rng.random(T) < 0.5, -1.5, 2.5)
    x[:, 6:10] = _field(rng, T, 4, kind="pr")
    x[:, 10] = np.minimum(x[:, 10], np.float32(290.0))
    x[:, 11] = np.round(x[:, 11] * 4) / 4                          # ~350 ties per value: > 2048 candidates, flagged
    x[:, 12] = np.round(x[:, 12])                                  # whole degrees
    x[:, 13] = x[:, 13] - 288
    x[: T - 1, 14] = np.nan
    x[:, 15] = np.where(rng.random(T) < 0.97, np.nan, x[:, 15])
    r16 = rng.random(T)
    x[:, 16] = np.where(r16 < 0.3, 0.0, np.where(r16 < 0.6, -0.0, x[:, 16] - 288))   # zeros of both signs
    x[:, C - 2] = np.nan
    q = osdba.equally_spaced_nodes(20)
    exp = osdba.quantile(x, q).astype(np.float32)
    out = K.quantile_series(dev, dev.to_device(x), q).get()
    np.testing.assert_array_equal(out, exp)
    xt = np.ascontiguousarray(x[:, :40].T)
    out2 = K.quantile_series(dev, dev.to_device(xt), q, time_axis=1).get()
    np.testing.assert_array_equal(out2, exp[:, :40])
    q40 = np.linspace(0.0, 1.0, 40)
    out3 = K.quantile_series(dev, dev.to_device(xt), q40, time_axis=1).get()
    np.testing.assert_array_equal(out3, osdba.quantile(x[:, :40], q40).astype(np.float32))


def test_quantile_series_two_pass_value_classes(dev, rng, monkeypatch):
    """The bins of select4.hip come from float differences against the window's ends (round 4): zeros of both signs,
    samples at +-FLT_MAX and +-inf (the window is clamped to finite ends), columns whose window has denormal width or
    overflows (scale 0), windows [x, nextafter(x)], a column of +inf.  Compared with the transposed column kernels
    (diagnostic switch: both are exact selections) and, for the finite columns, with the oracle."""
    T, C = 5000, 64 + 9
    x = _field(rng, T, C, nan_frac=0.001)
    fmax = np.finfo(np.float32).max
    x[:, 0] = np.where(rng.random(T) < 0.5, 0.0, -0.0)
    x[:, 1] = np.where(rng.random(T) < 0.4, -0.0, x[:, 1] - 288)
    x[:, 2] = np.where(rng.random(T) < 0.4, 0.0, x[:, 2] - 288)
    x[:, 3] = np.where(rng.random(T) < 0.1, -np.inf, x[:, 3])
    x[:, 4] = np.where(rng.random(T) < 0.1, np.inf, x[:, 4])
    x[:, 5] = np.where(rng.random(T) < 0.3, fmax, np.where(rng.random(T) < 0.3, -fmax, x[:, 5]))
    x[:, 6] = rng.choice(np.array([1e-45, 2.8e-45, 4.2e-45, 0.0], np.float32), T)
    x[:, 7] = np.where(rng.random(T) < 0.5, np.float32(1.0), np.nextafter(np.float32(1.0), np.float32(2.0)))
    x[:, 8] = np.inf
    x[:, 9] = np.where(rng.random(T) < 0.5, -fmax, fmax)
    x[:, 10] = (x[:, 10] - 288) * np.float32(1e-40)
    x[:, 11] = (x[:, 11] - 288) * np.float32(1e36)
    q = osdba.equally_spaced_nodes(20)
    xd = dev.to_device(x)
    a = K.quantile_series(dev, xd, q).get()
    monkeypatch.setenv("XH_DIAGNOSTICS", "1")
    monkeypatch.setenv("XH_SELECT_NOHIST", "1")
    b = K.quantile_series(dev, xd, q).get()
    np.testing.assert_array_equal(a, b)
    fin = np.ones(C, bool)
    fin[[3, 4, 8]] = False
    with np.errstate(all="ignore"):
        exp = osdba.quantile(x, q).astype(np.float32)
    np.testing.assert_array_equal(a[:, fin], exp[:, fin])


@pytest.mark.parametrize("T,C", [(365, 70001), (500, 33333), (800, 20011), (1500, 9001), (3650, 5003), (10950, 4801)])
def test_quantile_series_many_columns(dev, rng, T, C):
    """More columns than resident workgroups (grid-stride column loops, ragged last tiles of the staged time-major
    kernel, the padded transpose batches), every column checked against the oracle."""
    x = _field(rng, T, C, nan_frac=0.005)
    x[:, -1] = np.nan
    x[: T // 2, 17] = np.nan
    q = osdba.equally_spaced_nodes(20)
    exp = osdba.quantile(x, q)
    out = K.quantile_series(dev, dev.to_device(x), q).get()
    np.testing.assert_allclose(out, exp, rtol=RTOL, atol=0, equal_nan=True)
    out2 = K.quantile_series(dev, dev.to_device(np.ascontiguousarray(x.T)), q, time_axis=1).get()
    np.testing.assert_allclose(out2, exp, rtol=RTOL, atol=0, equal_nan=True)


@pytest.mark.parametrize("kind", ["+", "*"])
@pytest.mark.parametrize("interp", ["nearest", "linear"])
@pytest.mark.parametrize("extrap", ["constant", "nan"])
def test_eqm_train_adjust(dev, rng, kind, interp, extrap):
    T, C = 730, 150
    ref = _field(rng, T, C)
    hist = (_field(rng, T, C) + 1.5).astype(np.float32)
    sim = (_field(rng, T, C, nan_frac=0.01) + 2.0).astype(np.float32)
    hist[:, 0] = np.nan
    q = osdba.equally_spaced_nodes(20)
    af, hq = K.eqm_train(dev, dev.to_device(ref), dev.to_device(hist), q, kind)
    eaf, ehq = osdba.eqm_train(ref, hist, 20, kind)
    np.testing.assert_allclose(hq.get(), ehq, rtol=RTOL, equal_nan=True)
    np.testing.assert_allclose(af.get(), eaf, rtol=1e-5 if kind == "+" else RTOL, atol=1e-5 if kind == "+" else 0,
                               equal_nan=True)
    # adjust from the ORACLE's nodes so the comparison isolates the search/interp kernel (bit-exact expected)
    scen = K.eqm_adjust(dev, dev.to_device(sim), dev.to_device(eaf), dev.to_device(ehq), kind, interp, extrap).get()
    exp = osdba.eqm_adjust(sim, eaf, ehq, kind, interp, extrap)
    np.testing.assert_allclose(scen, exp, rtol=RTOL, atol=0, equal_nan=True)


@pytest.mark.parametrize("kind", ["+", "*"])
@pytest.mark.parametrize("extrap", ["constant", "nan"])
@pytest.mark.parametrize("nq", [20, 7, 32])
def test_eqm_adjust_cubic(dev, rng, kind, extrap, nq):
    """interp="cubic": scipy interp1d(kind="cubic") (not-a-knot spline) per cell, incl. NaN nodes dropped first."""
    T, C = 500, 90
    ref = _field(rng, T, C)
    hist = (_field(rng, T, C) + 1.5).astype(np.float32)
    sim = (_field(rng, T, C, nan_frac=0.01) + 2.0).astype(np.float32)
    eaf, ehq = osdba.eqm_train(ref, hist, nq, kind)
    eaf, ehq = eaf.astype(np.float32), ehq.astype(np.float32)
    eaf[1, 3] = np.nan           # an invalid node in one cell (dropped before the spline is built)
    ehq[nq - 2, 5] = np.nan
    scen = K.eqm_adjust(dev, dev.to_device(sim), dev.to_device(eaf), dev.to_device(ehq), kind, "cubic", extrap).get()
    exp = osdba.eqm_adjust(sim, eaf, ehq, kind, "cubic", extrap)
    np.testing.assert_allclose(scen, exp, rtol=2e-6, atol=0, equal_nan=True)
    assert np.isfinite(scen).mean() > 0.5


def test_synthetic_matches_oracle(dev):
    T, C = 400, 777
    base = osynth.seasonal_base(T)
    for kind, amp in ((0, 3.0), (1, 40.0 / 86400.0)):
        b = base if kind == 0 else np.zeros(T, np.float32)
        g = K.fill_synthetic(dev, T, C, kind, 42, b, amp, 0.3, 1000, cell0=5000).get()
        e = osynth.fill_synthetic(T, np.arange(5000, 5000 + C), kind, 42, b, amp, 0.3, 1000)
        np.testing.assert_array_equal(g, e)


@pytest.mark.parametrize("shape", [(130, 77), (128, 128), (300, 388), (257, 260), (5, 4), (8, 4), (1000, 132)])
def test_transpose(dev, rng, shape):
    """xh_transpose_f32: the 128 x 128 / 16-byte kernel (cols % 4 == 0: full and edge tiles) and the 64 x 64 one."""
    x = rng.normal(size=shape).astype(np.float32)
    np.testing.assert_array_equal(K.transpose(dev, dev.to_device(x)).get(), x.T)


@pytest.mark.parametrize("kind", ["+", "*"])
@pytest.mark.parametrize("interp", ["nearest", "linear"])
@pytest.mark.parametrize("extrap", ["constant", "nan"])
def test_eqm_precipitation_tied_nodes(dev, rng, kind, interp, extrap):
    """EQM on precipitation: most quantile nodes of hist are tied at exactly 0 (dry days), "*" factors are 0/0 = NaN
    (dropped) or x/0 = inf, and sim values sit exactly on the tied nodes.  Must follow scipy's interp1d on duplicate
    nodes (the oracle calls it), including the NaN / inf it produces."""
    T, C = 730, 120
    ref = _field(rng, T, C, kind="pr")
    hist = _field(rng, T, C, kind="pr")
    hist[:, :30] *= np.float32(0.0)                 # never wet: every node is 0
    ref[:, 30:60][ref[:, 30:60] < 2e-4] = 0.0       # drier ref: ref_q = 0 where hist_q > 0
    sim = _field(rng, T, C, kind="pr")
    sim[rng.random(sim.shape) < 0.01] = np.nan
    q = osdba.equally_spaced_nodes(20)
    af, hq = K.eqm_train(dev, dev.to_device(ref), dev.to_device(hist), q, kind)
    eaf, ehq = osdba.eqm_train(ref, hist, 20, kind)
    assert (ehq == 0).mean() > 0.3                  # the premise: tied zero nodes
    np.testing.assert_allclose(hq.get(), ehq, rtol=RTOL, equal_nan=True)
    np.testing.assert_allclose(af.get(), eaf, rtol=RTOL, atol=1e-12 if kind == "+" else 0, equal_nan=True)
    scen = K.eqm_adjust(dev, dev.to_device(sim), dev.to_device(eaf), dev.to_device(ehq), kind, interp, extrap).get()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        exp = osdba.eqm_adjust(sim, eaf, ehq, kind, interp, extrap)
    np.testing.assert_allclose(scen, exp, rtol=RTOL, atol=0, equal_nan=True)


@pytest.mark.parametrize("nq", [1, 2, 7, 10, 32, 50])
@pytest.mark.parametrize("interp", ["nearest", "linear"])
def test_eqm_adjust_node_counts(dev, rng, nq, interp):
    """Every register-resident node-count variant of the adjust kernel (10 / 20 / 32 / 64), with NaN nodes in some cells
    (compaction path of the linear search), one and two valid nodes, chunked time (few cells -> many time chunks)."""
    T, C = 413, 37
    ref = _field(rng, T, C)
    hist = (_field(rng, T, C) + 1.5).astype(np.float32)
    sim = (_field(rng, T, C, nan_frac=0.01) + 2.0).astype(np.float32)
    eaf, ehq = osdba.eqm_train(ref, hist, nq, "+")
    eaf, ehq = eaf.astype(np.float32), ehq.astype(np.float32)
    if nq >= 7:
        eaf[1, 3] = np.nan
        ehq[nq - 2, 5] = np.nan
        ehq[:, 7] = np.nan            # no valid node at all
        eaf[1:, 9] = np.nan           # one valid node
        eaf[2:, 11] = np.nan          # two valid nodes
    for extrap in ("constant", "nan"):
        scen = K.eqm_adjust(dev, dev.to_device(sim), dev.to_device(eaf), dev.to_device(ehq), "+", interp, extrap).get()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            exp = osdba.eqm_adjust(sim, eaf, ehq, "+", interp, extrap)
        np.testing.assert_allclose(scen, exp, rtol=RTOL, atol=0, equal_nan=True)
