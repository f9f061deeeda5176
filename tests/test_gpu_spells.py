"""GPU parity for the spell / hysteresis / season state machines (spell.hip) against the oracle."""

import numpy as np
import pytest

from oracle import generic as ogen
from oracle import run_length as orl
from oracle.timeutil import OTime
from xclim_amd import generic as xgen
from xclim_amd import run_length as xrl
from xclim_amd.timeaxis import TimeAxis

pytestmark = pytest.mark.gpu


def _pr(rng, T, C, nan_frac=0.0):
    x = np.where(rng.random((T, C)) < 0.45, rng.gamma(0.8, 8.0, (T, C)), 0.0).astype(np.float32)
    if nan_frac:
        x[rng.random((T, C)) < nan_frac] = np.nan
    return x


@pytest.mark.parametrize("window,red,op", [(3, "min", ">"), (3, "max", "<="), (4, "sum", ">="), (5, "mean", ">"), (2, "max", "<"),
                                           (1, None, ">"), (7, "min", ">=")])
@pytest.mark.parametrize("nan_frac", [0.0, 0.02])
def test_spell_mask(dev, rng, window, red, op, nan_frac):
    T, C = 150, 90
    x = _pr(rng, T, C, nan_frac)
    thr = {"sum": 12.0, "mean": 3.0}.get(red, 1.0)
    got = xgen.spell_mask(x, window, red, op, thr, device=dev)
    exp = ogen.spell_mask(x, window, red, op, thr)
    np.testing.assert_array_equal(got, exp)


def test_spell_mask_weights_and_gap(dev, rng):
    T, C = 120, 40
    x = _pr(rng, T, C)
    w = [0.5, 0.25, 0.25]
    np.testing.assert_array_equal(xgen.spell_mask(x, 3, "mean", ">", 2.0, weights=w, device=dev),
                                  ogen.spell_mask(x, 3, "mean", ">", 2.0, weights=w))
    for gap in (2, 3):
        np.testing.assert_array_equal(xgen.spell_mask(x, 2, "min", ">", 1.0, min_gap=gap, device=dev),
                                      ogen.spell_mask(x, 2, "min", ">", 1.0, min_gap=gap))
    with pytest.raises(ValueError, match="only supported"):
        xgen.spell_mask(x, 3, "max", ">", 2.0, weights=w, device=dev)
    with pytest.raises(ValueError, match="different length"):
        xgen.spell_mask(x, 4, "mean", ">", 2.0, weights=w, device=dev)


@pytest.mark.parametrize("window,red,reducer,before", [(3, "min", "max", True), (3, "sum", "count", True), (2, "max", "sum", False)])
def test_spell_length_statistics_general(dev, rng, window, red, reducer, before):
    T, C = 730, 60
    x = _pr(rng, T, C, 0.005)
    ta, ot = TimeAxis.daily("2001-01-01", T), OTime.standard("2001-01-01", T)
    op, thr = (">", 1.0) if red != "sum" else (">=", 10.0)
    got = xgen.spell_length_statistics(x, thr, window, red, op, reducer, ta, "YS", resample_before_rl=before, device=dev)
    exp = ogen.spell_length_statistics(x, thr, window, red, op, reducer, ot, "YS", resample_before_rl=before)
    np.testing.assert_array_equal(got, exp)
    got = xgen.spell_length_statistics(x, thr, window, red, op, reducer, ta, "YS", min_gap=2, resample_before_rl=before, device=dev)
    exp = ogen.spell_length_statistics(x, thr, window, red, op, reducer, ot, "YS", resample_before_rl=before, min_gap=2)
    np.testing.assert_array_equal(got, exp)


def test_reference_spell_length_statistics_answer(dev):
    """reference tests/test_generic.py:784-797 pattern: two spells of 34 and 4 days -> max 34."""
    x = np.zeros((365, 1), np.float32)
    x[10:44] = 5
    x[100:104] = 5
    ta = TimeAxis.daily("2001-01-01", 365)
    assert xgen.spell_length_statistics(x, 1.0, 3, "min", ">", "max", ta, "YS", device=dev)[0, 0] == 34
    assert xgen.spell_length_statistics(x, 1.0, 3, "min", ">", "count", ta, "YS", device=dev)[0, 0] == 2


@pytest.mark.parametrize("ws,wt", [(1, 1), (2, 3), (3, 2), (5, 5)])
def test_runs_with_holes(dev, rng, ws, wt):
    T, C = 200, 70
    a = rng.random((T, C)) < 0.5
    b = rng.random((T, C)) < 0.4
    np.testing.assert_array_equal(xrl.runs_with_holes(a, ws, b, wt, device=dev), orl.runs_with_holes(a, ws, b, wt))
    # identity (reference test_run_length.py:135-147): start = da, stop = ~da, windows 1
    np.testing.assert_array_equal(xrl.runs_with_holes(a, 1, ~a, 1, device=dev), a.astype(np.float32))


def test_keep_longest_run(dev, rng):
    T, C = 400, 50
    a = rng.random((T, C)) < 0.6
    a[:, 0] = False
    a[:, 1] = True
    ta, ot = TimeAxis.daily("2000-01-01", T), OTime.standard("2000-01-01", T)
    np.testing.assert_array_equal(xrl.keep_longest_run(a, device=dev), orl.keep_longest_run(a))
    np.testing.assert_array_equal(xrl.keep_longest_run(a, freq="MS", time=ta, device=dev), orl.keep_longest_run(a, ot, "MS"))
    # reference test_run_length.py:451-454
    m = np.array([1, 0, 0, 1, 1, 1, 0, 0, 1, 1, 1, 1], bool)[:, None]
    np.testing.assert_array_equal(xrl.keep_longest_run(m, device=dev)[:, 0], [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1])


@pytest.mark.parametrize("window", [1, 3, 5])
@pytest.mark.parametrize("mid_date", [None, "07-01", "02-10"])
def test_season(dev, rng, window, mid_date):
    T, C = 365 * 2 + 120, 80
    t = np.arange(T)[:, None]
    tas = 5 + 12 * np.sin(2 * np.pi * (t - 110) / 365) + rng.normal(0, 4, (T, C))
    cond = tas > 5
    cond[:, 0] = True
    cond[:, 1] = False
    ta, ot = TimeAxis.daily("2001-01-01", T), OTime.standard("2001-01-01", T)
    got = xrl.season(cond, window, mid_date, time=ta, freq="YS", device=dev)
    es, ee, el = orl.season_per_period(cond, window, mid_date, ot, "YS")
    np.testing.assert_array_equal(got["start"], es)
    np.testing.assert_array_equal(got["end"], ee)
    np.testing.assert_array_equal(got["length"], el)
    # whole-array form
    g1 = xrl.season(cond[:365], window, mid_date, time=ta.subset(slice(0, 365)), device=dev)
    s1, e1, l1 = orl.season(cond[:365], window, mid_date, ot.isel(slice(0, 365)))
    np.testing.assert_array_equal(g1["start"], s1)
    np.testing.assert_array_equal(g1["length"], l1)


def test_reference_season_answers(dev):
    """reference tests/test_run_length.py:484-502 (season_length 70 / 50 / 0) and :675-690 (start 140, end 150)."""
    ta = TimeAxis.daily("2000-01-01", 366)
    a = np.zeros((366, 1), bool)
    a[50:100] = True
    a[100:110] = False
    a[110:130] = True
    # mid_date inside the first run: season = first run of >=5 True .. first run of >= 5 False after mid
    out = xrl.season(a, 5, "03-01", time=ta, device=dev)
    assert out["start"][0] == 50 and out["end"][0] == 100 and out["length"][0] == 50
    b = np.zeros((366, 1), bool)
    assert xrl.season_length(b, 5, "07-01", time=ta, device=dev)[0] == 0
    c = np.zeros((366, 1), bool)
    c[140:150] = True
    o = xrl.season(c, 3, "05-25", time=ta, device=dev)  # doy 146 inside the run
    assert o["start"][0] == 140 and o["end"][0] == 150 and o["length"][0] == 10
    od = xrl.season(c, 3, "05-25", coord="dayofyear", time=ta, device=dev)
    assert od["start"][0] == 141 and od["end"][0] == 151


@pytest.mark.parametrize("window", [1, 2, 4])
@pytest.mark.parametrize("date", ["07-01", "03-15"])
def test_date_bounded_runs(dev, rng, window, date):
    """first_run_after_date / last_run_before_date / first_run_before_date / run_end_after_date (rl:1148-1331)."""
    T, C = 365 * 2 + 200, 70
    a = rng.random((T, C)) < 0.55
    a[:, 0] = True
    a[:, 1] = False
    ta, ot = TimeAxis.daily("2001-01-01", T), OTime.standard("2001-01-01", T)
    for xfn, ofn in ((xrl.first_run_after_date, orl.first_run_after_date), (xrl.last_run_before_date, orl.last_run_before_date),
                     (xrl.first_run_before_date, orl.first_run_before_date), (xrl.run_end_after_date, orl.run_end_after_date)):
        got = xfn(a, window, date, time=ta, freq="YS", device=dev)
        exp = orl.map_groups_fn(ofn, a, ot, "YS", window, date)
        np.testing.assert_array_equal(got, exp, err_msg=xfn.__name__)
    # the third "year" has only 200 days: 07-01 is inside (day 181), the group is still handled; an absent date -> NaN
    short = xrl.first_run_after_date(a[:100], 2, "07-01", time=ta.subset(slice(0, 100)), device=dev)
    assert np.isnan(short).all()
    g = xrl.first_run_after_date(a[:365], window, date, coord="dayofyear", time=ta.subset(slice(0, 365)), device=dev)
    e = orl.first_run_after_date(a[:365], window, date, ot.isel(slice(0, 365)))
    np.testing.assert_array_equal(g, e + 1)


@pytest.mark.parametrize("window", [1, 3])
def test_windowed_max_run_sum(dev, rng, window):
    T, C = 300, 60
    x = _pr(rng, T, C)
    ta, ot = TimeAxis.daily("2002-01-01", T), OTime.standard("2002-01-01", T)
    got = xrl.windowed_max_run_sum(x, window, device=dev)
    np.testing.assert_allclose(got, orl.windowed_max_run_sum(x, window), rtol=1e-6)
    x[:, 7] = 0.0                      # no run at all
    x[20:140, 8] = 2.5e-4              # one run across five months: it counts for the month of its first day only
    for freq in ("MS", "QS-DEC", "YS"):
        # resample before (runs cut at the period edges): rl.resample_and_rl(..., resample_before_rl=True)
        exp = orl.resample_and_rl(x, True, orl.windowed_max_run_sum, window, time=ot, freq=freq)
        got = xrl.resample_and_rl(x, True, xrl.windowed_max_run_sum, window, freq=freq, time=ta, device=dev)
        np.testing.assert_allclose(got, exp, rtol=1e-6)
        np.testing.assert_allclose(xrl.windowed_max_run_sum(x, window, freq=freq, time=ta, device=dev, cut=True), exp, rtol=1e-6)
        # the reference's own `freq` argument resamples AFTER: cumsum and run lengths cross the period edges
        exp = orl.windowed_max_run_sum(x, window, ot, freq)
        got = xrl.windowed_max_run_sum(x, window, freq=freq, time=ta, device=dev)
        np.testing.assert_allclose(got, exp, rtol=1e-6)
        np.testing.assert_allclose(xrl.resample_and_rl(x, False, xrl.windowed_max_run_sum, window, freq=freq, time=ta, device=dev),
                                   exp, rtol=1e-6)
        # index="last": the run counts for the period of its LAST day, its sum is accumulated forward (bit-exact order)
        exp = orl.windowed_max_run_sum(x, window, ot, freq, index="last")
        got = xrl.windowed_max_run_sum(x, window, freq=freq, time=ta, device=dev, index="last")
        np.testing.assert_array_equal(got, exp)
    np.testing.assert_array_equal(xrl.windowed_max_run_sum(x, window, device=dev, index="last"),
                                  orl.windowed_max_run_sum(x, window, index="last"))
    f = np.zeros((50, 1), np.float32)
    f[4:6] = 5
    f[25:30] = 5
    f[35:45] = 5
    assert xrl.windowed_max_run_sum(f, 3, device=dev)[0] == 50  # reference test_run_length.py:373-381


def _runs_mask(rng, T, C, p_on=0.55, nan_frac=0.0):
    m = (rng.random((T, C)) < p_on).astype(np.float32)
    # lengthen the runs a little so that windows > 1 find something
    m = np.maximum(m, np.roll(m, 1, axis=0) * (rng.random((T, C)) < 0.5))
    if nan_frac:
        m[rng.random((T, C)) < nan_frac] = np.nan
    return m.astype(np.float32)


@pytest.mark.gpu
def test_run_bounds(dev, rng):
    """rl:745-802: start / end indices of every run, NaN padded to the busiest cell."""
    m = _runs_mask(rng, 200, 37).astype(bool)
    m[:, 0] = False  # no run at all
    m[:, 1] = True   # one run that never ends
    m[-1, 2] = True
    got = xrl.run_bounds(m, device=dev)
    ref = orl.run_bounds(m)
    assert got.shape == ref.shape
    np.testing.assert_array_equal(got, ref)
    # hand-checked example
    one = np.array([0, 1, 1, 0, 1, 0, 0, 1, 1, 1], bool)[:, None]
    np.testing.assert_array_equal(xrl.run_bounds(one, device=dev)[:, :, 0], [[1, 4, 7], [3, 5, np.nan]])


@pytest.mark.gpu
@pytest.mark.parametrize("window,window_stop", [(1, 1), (2, 1), (3, 2)])
@pytest.mark.parametrize("freq", [None, "MS"])
def test_find_events(dev, rng, window, window_stop, freq):
    """rl:1760-1901: event table (length, effective length, start, sum) incl. the sum-stops-at-NaN arithmetic."""
    T, C = 150, 29
    cond = _runs_mask(rng, T, C).astype(bool)
    stop = (rng.random((T, C)) < 0.4)
    data = rng.gamma(2.0, 3.0, (T, C)).astype(np.float32)
    data[rng.random((T, C)) < 0.05] = np.nan
    ta, ot = TimeAxis.daily("2001-01-01", T, "standard"), OTime.standard("2001-01-01", T)
    for cs in (None, stop):
        got = xrl.find_events(cond, window, cs, window_stop, data=data, freq=freq, time=ta, device=dev)
        ref = orl.find_events(cond, window, cs, window_stop, data=data, freq=freq, time=ot)
        assert set(got) == set(ref)
        for k in ref:
            assert got[k].shape == ref[k].shape, k
            if k == "event_sum":
                np.testing.assert_allclose(got[k], ref[k], rtol=1e-6, equal_nan=True)
            else:
                np.testing.assert_array_equal(got[k], ref[k], err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("window,op,thresh", [(3, ">", None), (2, ">", 1.0), (1, "==", 2.0), (4, "<=", 3.0), (2, "!=", 0.0)])
def test_suspicious_run(dev, rng, window, op, thresh):
    x = rng.integers(0, 4, (300, 23)).astype(np.float32)
    x[rng.random(x.shape) < 0.05] = np.nan
    x[100:140, 3] = 2.0
    got = xrl.suspicious_run(x, window=window, op=op, thresh=thresh, device=dev)
    np.testing.assert_array_equal(got, orl.suspicious_run(x, window, op, thresh))


@pytest.mark.gpu
@pytest.mark.parametrize("freq", [None, "YS"])
def test_rle_statistics_quantile(dev, rng, freq):
    """rl:318-327 with "qNN" reducers: quantile (linear) of the run lengths >= window, 0 without any."""
    T, C = 730, 41
    m = _runs_mask(rng, T, C, nan_frac=0.01)
    m[:, 0] = 0
    ta, ot = TimeAxis.daily("2001-01-01", T, "standard"), OTime.standard("2001-01-01", T)
    for red, window in (("q90", 1), ("q10", 2), ("q50", 3)):
        got = xrl.rle_statistics(m, red, window, freq=freq, time=ta, device=dev)
        ref = orl.rle_statistics(m, red, window, time=ot, freq=freq)
        np.testing.assert_allclose(got, ref, rtol=1e-6)


@pytest.mark.gpu
def test_reference_run_quantile_answers(dev):
    """tests/test_run_length.py:270-296 of the reference: q90 = 299.6, q10 = 64.4 on the synthetic run pattern."""
    values = np.zeros(365, np.float32)
    values[1:11] = 1
    values[20:350] = 1
    values[355:] = 1  # runs of 10, 330, 10
    got90 = xrl.rle_statistics(values[:, None], "q90", 1, device=dev)
    got10 = xrl.rle_statistics(values[:, None], "q10", 1, device=dev)
    np.testing.assert_allclose(got90, np.quantile([10, 330, 10], 0.9), rtol=1e-6)
    np.testing.assert_allclose(got10, np.quantile([10, 330, 10], 0.1), rtol=1e-6)


@pytest.mark.gpu
def test_1d_variants_and_season_end(dev, rng):
    """rl:1334-1618: the 1-D ("ufunc") spellings give the N-D results."""
    v = (rng.random(400) < 0.6)
    assert xrl.statistics_run_1d(v, "max", 2, device=dev) == orl.statistics_run_1d(v, "max", 2)
    assert xrl.windowed_run_count_1d(v, 3, device=dev) == orl.windowed_run_count_1d(v, 3)
    assert xrl.windowed_run_events_1d(v, 3, device=dev) == orl.windowed_run_events_1d(v, 3)
    f = xrl.first_run_1d(v, 4, device=dev)
    r = orl.first_run_1d(v, 4)
    assert (np.isnan(f) and np.isnan(r)) or f == r
    vals, lens, pos = xrl.rle_1d(np.array([1, 2, 2, 3, 3, 3]))
    np.testing.assert_array_equal(lens, [1, 2, 3])
    np.testing.assert_array_equal(pos, [0, 1, 3])
    m = _runs_mask(rng, 365, 11).astype(bool)
    ta = TimeAxis.daily("2001-01-01", 365, "standard")
    np.testing.assert_array_equal(xrl.season_end(m, 3, "07-01", time=ta, device=dev),
                                  xrl.season(m, 3, "07-01", time=ta, device=dev)["end"])


@pytest.mark.gpu
@pytest.mark.parametrize("window,red,op", [(1, "min", ">"), (3, "min", ">="), (3, "max", "<"), (4, "mean", ">"), (2, "sum", "<="),
                                           (3, "max", "<=")])
@pytest.mark.parametrize("var_reducer", ["all", "any"])
def test_spell_mask_two_variables(dev, rng, window, red, op, var_reducer):
    """gen:434-540 with a list of variables, incl. the daily-combination fast path of gen:503-518."""
    a = _pr(rng, 300, 31, nan_frac=0.02)
    b = _pr(rng, 300, 31, nan_frac=0.02) * 1.5
    th = [2.0, 3.5] if red != "sum" else [5.0, 8.0]
    got = xgen.spell_mask([a, b], window, red, op, th, var_reducer=var_reducer, device=dev)
    ref = ogen.spell_mask([a, b], window, red, op, th, var_reducer=var_reducer)
    np.testing.assert_array_equal(got, ref)
    got2 = xgen.spell_mask([a, b], window, red, op, th, min_gap=3, var_reducer=var_reducer, device=dev)
    np.testing.assert_array_equal(got2, ogen.spell_mask([a, b], window, red, op, th, min_gap=3, var_reducer=var_reducer))


@pytest.mark.gpu
def test_bivariate_spell_length_statistics_and_thresholded_events(dev, rng):
    T = 500
    pr = _pr(rng, T, 23, nan_frac=0.01)
    tas = (rng.normal(2.0, 4.0, (T, 23))).astype(np.float32)
    ta, ot = TimeAxis.daily("2001-03-01", T, "standard"), OTime.standard("2001-03-01", T)
    for window, red in ((1, "min"), (3, "min"), (3, "mean")):
        for sr in ("max", "sum", "count"):
            got = xgen.bivariate_spell_length_statistics(pr, 1.0, tas, 0.0, window, red, ">=", sr, ta, "YS", device=dev)
            ref = ogen.spell_length_statistics([pr, tas], [1.0, 0.0], window, red, ">=", sr, ot, "YS")
            np.testing.assert_array_equal(got, ref)
    both = xgen.bivariate_spell_length_statistics(pr, 1.0, tas, 0.0, 2, "min", ">=", ("max", "count"), ta, "MS", device=dev)
    assert isinstance(both, tuple) and len(both) == 2
    with pytest.raises(ValueError):
        xgen.spell_mask([pr, tas], 2, "min", ">", 1.0, device=dev)
    # thresholded_events (gen:1739-1804) = find_events on compare masks
    for kw in ({}, {"thresh_stop": 0.5}, {"op_stop": "<", "thresh_stop": 0.2}):
        got = xgen.thresholded_events(pr, 1.0, ">=", 2, window_stop=2, time=ta, device=dev, **kw)
        start = ogen.compare(pr, ">=", 1.0)
        if not kw:
            stop = ~start
        elif "op_stop" in kw:
            stop = ogen.compare(pr, "<", 0.2)
        else:
            stop = ~ogen.compare(pr, ">=", 0.5)
        ref = orl.find_events(start, 2, stop, 2, data=pr)
        for k in ref:
            np.testing.assert_allclose(got[k], ref[k], rtol=1e-6, equal_nan=True, err_msg=k)


@pytest.mark.gpu
def test_runs_with_holes_long_windows_and_pool(dev, rng):
    """windows > 64 take the two-pass kernel (the single-pass one keeps its marks in 64-bit shift registers); both must
    agree with the oracle.  Also: freed device buffers are re-used by the pooled allocator of the Python Device."""
    T, C = 900, 19
    a = (rng.random((T, C)) < 0.93)
    b = (rng.random((T, C)) < 0.9)
    for ws, wt in ((70, 3), (5, 66), (64, 64), (65, 1)):
        got = xrl.runs_with_holes(a, ws, b, wt, device=dev)
        np.testing.assert_array_equal(got, orl.runs_with_holes(a, ws, b, wt))
    x = dev.empty((1234, 7), np.float32)
    ptr = x.ptr
    x.free()
    y = dev.empty((1234, 7), np.float32)
    assert y.ptr == ptr  # same size -> the pooled buffer comes back
    y.free()
    dev.trim()
    z = dev.empty((1234, 7), np.float32)
    z.free()


@pytest.mark.parametrize("T,C", [(730, 37), (365, 1024), (1461, 5)])
@pytest.mark.parametrize("op", [">", "<=", "!="])
def test_run_stats_doy_fused(dev, rng, T, C, op):
    """xh_run_stats_doy == compare against the broadcast per-doy table, then run statistics cut at the period edges
    (rl.resample_and_rl(..., resample_before_rl=True)); every reducer, ragged periods, an empty period, NaN in x and in
    the table."""
    from oracle import run_length as orl
    from xclim_amd import kernels as K

    x = rng.normal(0, 1, (T, C)).astype(np.float32)
    x += np.repeat(rng.normal(0, 1.5, (T // 5 + 1, C)), 5, axis=0)[:T].astype(np.float32)
    x[rng.random(x.shape) < 0.01] = np.nan
    D = 366
    table = rng.normal(0, 0.5, (D, C))
    table[rng.random(table.shape) < 0.01] = np.nan
    tidx = (np.arange(T) * 7 % D).astype(np.int32)
    seg = np.array([0, 31, 31, 59, 200, T], dtype=np.int64)
    dx, dt = dev.to_device(x), dev.to_device(table)
    with np.errstate(invalid="ignore"):
        cond = {">": np.greater, "<=": np.less_equal, "!=": np.not_equal}[op](x.astype(np.float64), table[tidx])
    for stat in ("max", "min", "sum", "count", "mean", "std"):
        for window in (1, 3):
            got, val = K.run_stats_doy(dev, dx, op, dt, tidx, stat, window, seg)
            exp = np.stack([orl.rle_statistics(cond[a:b], stat, window) if b > a else np.zeros(C)
                            for a, b in zip(seg[:-1], seg[1:])])
            np.testing.assert_allclose(got.get(), exp, rtol=1e-6, err_msg=f"{stat} w{window}")
            two, _ = K.run_stats(dev, K.compare_doy(dev, dx, op, dt, tidx), stat, window, seg, cut=True, want_valid=False)
            np.testing.assert_array_equal(got.get(), two.get())
    np.testing.assert_array_equal(val.get(), np.stack([(~np.isnan(x[a:b])).sum(axis=0) for a, b in zip(seg[:-1], seg[1:])]))
    with pytest.raises(Exception):
        K.run_stats_doy(dev, dx, op, dt, tidx + D, "sum", 1, seg)


@pytest.mark.parametrize("reducer", ["max", "min", "mean", "sum"])
@pytest.mark.parametrize("op", [">", "<=", "=="])
def test_spell_length(dev, rng, reducer, op):
    """generic.spell_length (gen:1204-1252): run statistics of the compared series, cut at the period edges."""
    T = 800
    x = np.round(rng.normal(0, 2.0, (T, 6, 7)) + np.repeat(rng.normal(0, 2.0, (T // 4, 6, 7)), 4, axis=0)).astype(np.float32)
    x[rng.random(x.shape) < 0.01] = np.nan
    ta, ot = TimeAxis.daily("2001-03-01", T), OTime.standard("2001-03-01", T)
    for freq in ("YS", "MS", "QS-DEC"):
        got = xgen.spell_length(x, 1.0, reducer, ta, freq, op, device=dev)
        ref = ogen.spell_length(x, np.float32(1.0), reducer, ot, freq, op)
        np.testing.assert_allclose(got, ref, rtol=1e-6)
    with pytest.raises(ValueError):
        xgen.spell_length(x, 1.0, "std", ta, "YS", op, device=dev)


@pytest.mark.parametrize("coord", ["dayofyear", "month", "day", "year"])
def test_boundary_run_coord(dev, rng, coord):
    """first_run / last_run / first_run_after_date with coord=<datetime field> (rl:586-597: the index is looked up in
    da.time.dt.<coord>), whole series and per period."""
    T = 900
    x = (rng.random((T, 4, 5)) < 0.35).astype(np.float32)
    x[:, 0, 0] = 0.0  # no run -> NaN stays NaN
    ta, ot = TimeAxis.daily("2001-02-10", T), OTime.standard("2001-02-10", T)
    table = {"dayofyear": ot.doy, "month": ot.month, "day": ot.day, "year": ot.year}[coord]

    def lookup(idx, offsets):
        out = np.full(idx.shape, np.nan)
        for p, off in enumerate(offsets):
            ok = ~np.isnan(idx[p])
            out[p][ok] = table[off + idx[p][ok].astype(int)]
        return out

    for window in (1, 3):
        for f, of in ((xrl.first_run, orl.first_run), (xrl.last_run, orl.last_run)):
            got = f(x, window, coord=coord, time=ta, device=dev)
            exp = lookup(of(x > 0, window)[None], [0])[0]
            np.testing.assert_array_equal(got, exp)
            seg, _ = ta.segments("YS")
            got = f(x, window, freq="YS", coord=coord, time=ta, device=dev)
            exp = lookup(of(x > 0, window, ot, "YS"), seg[:-1])
            np.testing.assert_array_equal(got, exp)
    assert np.isnan(xrl.first_run(x, 3, coord=coord, time=ta, device=dev)[0, 0])
    with pytest.raises(NotImplementedError):
        xrl.first_run(x, 3, coord="hour", time=ta, device=dev)
    with pytest.raises(ValueError):
        xrl.first_run(x, 3, coord=coord, device=dev)


def test_boundary_run_coord_true_returns_dates(dev):
    """coord=True: the time coordinate itself (rl:586-593 via utils.lazy_indexing).  Known answers of the reference:
    tests/test_run_length.py:314-353 (first_run -> 2000-01-31, per month -> 2000-01-01 / 2000-02-01),
    :384-402 (last_run -> 2000-02-09); no run -> NaT."""
    t = np.zeros((60, 2), np.float32)
    t[30:40] = 2
    ta = TimeAxis.daily("2000-01-01", 60)
    out = xrl.first_run(t > 0, window=1, coord=True, time=ta, device=dev)
    assert out.dtype == np.dtype("datetime64[ns]")
    np.testing.assert_array_equal(out, np.array(["2000-01-31"] * 2, dtype="datetime64[ns]"))
    np.testing.assert_array_equal(xrl.last_run(t > 0, window=1, coord=True, time=ta, device=dev),
                                  np.array(["2000-02-09"] * 2, dtype="datetime64[ns]"))
    t[0] = 2
    out = xrl.first_run(t > 0, window=1, coord=True, freq="MS", time=ta, device=dev)
    np.testing.assert_array_equal(out, np.array([["2000-01-01"] * 2, ["2000-02-01"] * 2], dtype="datetime64[ns]"))
    t[:, 1] = 0
    out = xrl.first_run(t > 0, window=3, coord=True, time=ta, device=dev)
    assert out[0] == np.datetime64("2000-01-31") and np.isnat(out[1])
    # season start / end as dates; calendars without datetime64 (360_day has Feb 30) give ISO strings / None
    c = np.zeros((720, 1), np.float32)
    c[40:400] = 1
    t360 = TimeAxis.daily("2001-01-01", 720, "360_day")
    s = xrl.season(c > 0, 3, coord=True, time=t360, device=dev)
    assert s["start"][0] == "2001-02-11" and s["end"][0] == "2002-02-11"
    assert xrl.first_run(np.zeros((720, 1), np.float32), 3, coord=True, time=t360, device=dev)[0] is None


def test_device_mask_chaining_and_dtype_guard(dev, rng):
    """compare(..., keep=True) hands a float32 device mask to the run-length mirrors (the drop-in chain
    cond = compare(...); rl.rle_statistics(cond, ...) without leaving the device); a uint8 / float64 device buffer is
    refused instead of being reinterpreted as float32."""
    from xclim_amd import kernels as K

    T = 400
    x = rng.normal(0, 1, (T, 6, 5)).astype(np.float32)
    x[rng.random(x.shape) < 0.02] = np.nan
    ta, ot = TimeAxis.daily("2001-01-01", T), OTime.standard("2001-01-01", T)
    cond = xgen.compare(x, ">", 0.25, device=dev, keep=True)
    assert cond.dtype == np.float32
    got = xrl.rle_statistics(cond.reshape(T, 6, 5), "max", 2, freq="MS", time=ta, device=dev)
    exp = orl.rle_statistics(ogen.compare(x, ">", np.float32(0.25)), "max", 2, ot, "MS")
    np.testing.assert_array_equal(got, exp)
    np.testing.assert_array_equal(xgen.compare(x, ">", 0.25, device=dev), ogen.compare(x, ">", np.float32(0.25)))
    u8 = K.compare_map(dev, dev.to_device(x.reshape(T, -1)), ">", 0.25, "mask")
    with pytest.raises(TypeError, match="float32"):
        xrl.rle_statistics(u8, "max", 2, device=dev)
    x64 = dev.to_device(x.astype(np.float64).reshape(T, -1))
    with pytest.raises(TypeError, match="float32"):
        xgen.spell_length_statistics(x64, 0.0, 1, None, ">", "max", ta, "YS", device=dev)
    # (threshold_count has a float64 kernel since round 3: the float64 device field is taken as it is)
    np.testing.assert_array_equal(xgen.threshold_count(x64, ">", 0.0, ta, "YS", device=dev).reshape(-1, 6, 5),
                                  ogen.threshold_count(x.astype(np.float64), ">", 0.0, ot, "YS"))


@pytest.mark.parametrize("shape", [(37, 5, 7), (365, 1003), (1, 3), (50,)])
def test_bool_masks_upload_as_bytes(dev, rng, shape):
    """numpy bool / uint8 masks go to the device as bytes and are widened there (xh_mask_u8_to_f32): same results as the
    float32 mask, any size (tail not a multiple of 4)."""
    from xclim_amd import kernels as K

    m = rng.random(shape) < 0.4
    f = m.astype(np.float32)
    for fn, args in ((xrl.rle, ()), (xrl.longest_run, ()), (xrl.windowed_run_count, (2,)), (xrl.first_run, (2,))):
        a = fn(f, *args, device=dev)
        np.testing.assert_array_equal(fn(m, *args, device=dev), a)
        np.testing.assert_array_equal(fn(m.astype(np.uint8), *args, device=dev), a)
    d8 = dev.to_device(m.reshape(shape[0], -1).view(np.uint8))
    np.testing.assert_array_equal(K.mask_to_f32(dev, d8).get(), f.reshape(shape[0], -1))
    with pytest.raises(TypeError):
        K.mask_to_f32(dev, dev.to_device(f.reshape(shape[0], -1)))


@pytest.mark.parametrize("before", [True, False])
@pytest.mark.parametrize("window", [1, 3])
def test_spell_length_statistics_indexer(dev, rng, before, window):
    """gen:543-585: ``**indexer`` masks the spell MASK (NaN outside the selection), so the run-length statistics run on a
    mask with NaN steps — the rle visibility quirk included (a run whose outer neighbour is NaN loses its length)."""
    T = 1096
    x = rng.gamma(0.8, 4.0, (T, 5, 6)).astype(np.float32)
    x[rng.random(x.shape) < 0.55] = 0.0
    x[rng.random(x.shape) < 0.003] = np.nan
    ta, ot = TimeAxis.daily("2001-01-01", T), OTime.standard("2001-01-01", T)
    for indexer in (dict(season="JJA"), dict(month=[4, 5, 6, 7]), dict(date_bounds=("11-01", "03-15"))):
        for red in ("max", "sum", "count"):
            for freq in ("YS", "QS-DEC"):
                got = xgen.spell_length_statistics(x, 1.0, window, "sum" if window > 1 else None, "<", red, ta, freq,
                                                   resample_before_rl=before, device=dev, **indexer)
                ref = ogen.spell_length_statistics(x, np.float32(1.0), window, "sum" if window > 1 else None, "<", red, ot, freq,
                                                   resample_before_rl=before, **indexer)
                np.testing.assert_array_equal(got, ref, err_msg=f"{indexer} {red} {freq}")


def test_rolling_and_bivariate_indexer(dev, rng):
    """``**indexer`` of select_rolling_resample_op (applied to the rolled series, gen:174) and of
    bivariate_spell_length_statistics (applied to the combined spell mask)."""
    from oracle import calendar as ocal

    T = 800
    x = rng.normal(10, 3, (T, 4, 5)).astype(np.float32)
    y = (x + rng.normal(0, 2, x.shape)).astype(np.float32)
    ta, ot = TimeAxis.daily("2001-01-01", T), OTime.standard("2001-01-01", T)
    got = xgen.select_rolling_resample_op(x, "max", 5, ta, True, "mean", "YS", device=dev, season="MAM")
    ref = ogen.select_resample_op(ocal.select_time(ogen.rolling(x, 5, "mean", True), ot, season="MAM"), "max", ot, "YS")
    np.testing.assert_allclose(got, ref, rtol=1e-6, equal_nan=True)
    got = xgen.bivariate_spell_length_statistics(x, 9.0, y, 9.0, 3, "min", ">=", "max", ta, "YS", device=dev, month=[5, 6, 7])
    ref = ogen.spell_length_statistics([x, y], [np.float32(9.0), np.float32(9.0)], 3, "min", ">=", "max", ot, "YS", month=[5, 6, 7])
    np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("T,C", [(731, 33), (400, 1024)])
@pytest.mark.parametrize("win_reducer", ["sum", "mean", "min", "max", "wmean"])
def test_spell_run_stats_fused_equals_two_step(dev, rng, T, C, win_reducer):
    """xh_spell_run_stats == xh_spell_mask followed by xh_run_stats (window 1, cut at the periods), bit for bit: every
    window reducer (incl. weights), windows 2..8, every run statistic, ragged / empty periods, NaN steps; windows > 8 are
    declined (None) so that callers take the two-step path."""
    from xclim_amd import kernels as K

    x = rng.gamma(0.8, 3.0, (T, C)).astype(np.float32)
    x[rng.random(x.shape) < 0.5] = 0.0
    x[rng.random(x.shape) < 0.01] = np.nan
    dx = dev.to_device(x)
    seg = np.array([0, 31, 31, 59, 200, T], dtype=np.int64)
    for window in (2, 3, 5, 8):
        weights = (rng.random(window) + 0.5).astype(np.float32) if win_reducer == "wmean" else None
        red = "mean" if win_reducer == "wmean" else win_reducer
        for op, thr in ((">=", 2.0), ("<", 1.0)):
            mask = K.spell_mask(dev, dx, window, red, op, thr, weights)
            for stat in ("max", "min", "sum", "count", "mean", "std"):
                two, _ = K.run_stats(dev, mask, stat, 1, seg, cut=True, want_valid=False)
                fused, val = K.spell_run_stats(dev, dx, window, red, op, thr, stat, seg, weights=weights)
                np.testing.assert_array_equal(fused.get(), two.get(), err_msg=f"w{window} {op} {stat}")
            np.testing.assert_array_equal(val.get(), np.stack([(~np.isnan(x[a:b])).sum(axis=0) for a, b in zip(seg[:-1], seg[1:])]))
    assert K.spell_run_stats(dev, dx, 9, "sum", ">", 1.0, "max", seg) is None


def test_use_ufunc_dispatch_by_grid_size(dev, rng):
    """rl:33-78: the same NaN-carrying mask gives the 1-D semantics under 9000 cells and the N-D semantics (rle quirk) at
    9000 cells and more, with the default ``ufunc_1dim="from_context"`` — mirrors and oracle dispatch alike, and the two
    semantics really differ on this input."""
    T = 90
    small = (rng.random((T, 40, 50)) < 0.6).astype(np.float32)      # 2000 cells
    small[rng.random(small.shape) < 0.05] = np.nan
    big = np.tile(small, (1, 1, 5))[:, :, :225]                      # 40 x 225 = 9000 cells, same columns repeated
    assert xrl.use_ufunc("from_context", small) and not xrl.use_ufunc("from_context", big)
    assert not xrl.use_ufunc("from_context", small, freq="YS") and not xrl.use_ufunc("from_context", small, index="last")
    for red in ("max", "sum", "mean"):
        gs = xrl.rle_statistics(small, red, 2, device=dev)
        gb = xrl.rle_statistics(big, red, 2, device=dev)
        np.testing.assert_allclose(gs, orl.rle_statistics(small, red, 2), rtol=1e-6, equal_nan=True)
        np.testing.assert_allclose(gb, orl.rle_statistics(big, red, 2), rtol=1e-6, equal_nan=True)
        np.testing.assert_allclose(gb, orl.rle_statistics(big, red, 2, ufunc_1dim=False), rtol=1e-6, equal_nan=True)
        np.testing.assert_allclose(xrl.rle_statistics(small, red, 2, device=dev, ufunc_1dim=False),
                                   orl.rle_statistics(small, red, 2, ufunc_1dim=False), rtol=1e-6, equal_nan=True)
    assert not np.array_equal(xrl.rle_statistics(small, "sum", 2, device=dev),
                              xrl.rle_statistics(small, "sum", 2, device=dev, ufunc_1dim=False))
    np.testing.assert_array_equal(xrl.windowed_run_count(small, 2, device=dev), orl.windowed_run_count(small, 2))
    np.testing.assert_array_equal(xrl.windowed_run_events(big, 2, device=dev), orl.windowed_run_events(big, 2))
    with pytest.raises(ValueError, match="1d method"):
        xrl.rle_statistics(small, "max", 2, freq="YS", ufunc_1dim=True, time=TimeAxis.daily("2001-01-01", T), device=dev)


def test_more_run_length_reference_known_answers(dev):
    """tests/test_run_length.py:135-160, 427-434, 451-454, 472-560 through the HIP path (incl. coord="dayofyear")."""
    from tests.test_oracle_reference_answers import _FIRST_RUN_AFTER, _RUN_END_AFTER, _SEASON_LENGTH, _rwh_series

    values, expected = _rwh_series()
    np.testing.assert_array_equal(xrl.runs_with_holes((values == 1)[:, None], 1, (values == 0)[:, None], 3, device=dev)[:, 0], expected)
    ident = np.zeros((365, 4, 4), np.float32)
    ident[1:11] = 1
    np.testing.assert_array_equal(xrl.runs_with_holes(ident != 0, 1, ident == 0, 1, device=dev), ident)
    runs = np.array([0, 1, 1, 1, 0, 0, 1, 1, 1, 0], dtype=bool)[:, None]
    np.testing.assert_array_equal(xrl.keep_longest_run(runs, device=dev)[:, 0], [0, 1, 1, 1, 0, 0, 0, 0, 0, 0])
    np.testing.assert_array_equal(xrl.run_bounds(runs, device=dev)[:, :, 0], [[1, 6], [4, 9]])
    ta = TimeAxis.daily("2000-01-01", 360)
    for date, end, exp in _SEASON_LENGTH:
        t = np.zeros((360, 2), np.float32)
        t[140:end] = 1
        np.testing.assert_array_equal(xrl.season_length(t == 1, 1, date, time=ta, device=dev), [exp, exp])
    for date, end, exp in _RUN_END_AFTER:
        t = np.zeros((360, 2), np.float32)
        t[140:end] = 1
        np.testing.assert_array_equal(xrl.run_end_after_date(t == 1, 1, date, time=ta, device=dev), [exp, exp])
        doy = xrl.run_end_after_date(t == 1, 1, date, coord="dayofyear", time=ta, device=dev)
        np.testing.assert_array_equal(doy, [exp + 1, exp + 1])  # 2000-01-01 is day 1: 211 / 191 / NaN / 306
    ta5 = TimeAxis.daily("2000-01-01", 365)
    for date, beg, exp in _FIRST_RUN_AFTER:
        t = np.zeros((365, 2), np.float32)
        if beg:
            t[beg:] = 1
        np.testing.assert_array_equal(xrl.first_run_after_date(t == 1, 1, date, time=ta5, device=dev), [exp, exp])


def test_season_and_find_events_reference_known_answers(dev):
    """tests/test_run_length.py:675-730 through the HIP path (xh_season, xh_runs_with_holes + xh_run_events)."""
    from tests.test_oracle_reference_answers import _FIND_EVENTS_COND

    t = np.zeros((360, 2), np.float32)
    t[140:150] = 1
    out = xrl.season(t >= 1, 2, time=TimeAxis.daily("2000-01-01", 360), device=dev)
    np.testing.assert_array_equal(out["start"], [140, 140])
    np.testing.assert_array_equal(out["end"], [150, 150])
    np.testing.assert_array_equal(out["length"], [10, 10])
    ev = xrl.find_events(_FIND_EVENTS_COND, 1, device=dev)
    exp = np.pad(np.array([[4, np.nan], [2, 4], [4, 1]]), [(0, 0), (0, 4)], constant_values=np.nan).T
    np.testing.assert_equal(ev["event_length"], exp)
    np.testing.assert_equal(ev["event_start"][0], [3, 2, 1])
    ev = xrl.find_events(_FIND_EVENTS_COND, 2, None, 3, device=dev)
    np.testing.assert_equal(ev["event_length"], np.pad(np.array([[4.0], [9.0], [7.0]]), [(0, 0), (0, 2)], constant_values=np.nan).T)


def test_spell_mask_reference_known_answers(dev):
    """tests/test_generic.py:702-766 through the HIP path (xh_spell_mask / xh_spell_mask_multi, per-site thresholds),
    with the reference's argument errors."""
    from tests.test_oracle_reference_answers import _SM_D1, _SM_D2, _SM_MULTI, _SM_SINGLE

    for (w, red, op, thr, weights), exp in _SM_SINGLE:
        got = xgen.spell_mask(_SM_D1[:, None], w, red, op, thr, weights=weights, device=dev)[:, 0]
        np.testing.assert_array_equal(got, np.array(exp, bool), err_msg=f"{w} {red} {op}")
    for (w, red, op, thr, weights, vr), exp in _SM_MULTI:
        got = xgen.spell_mask([_SM_D1[:, None], _SM_D2[:, None]], w, red, op, thr, weights=weights, var_reducer=vr, device=dev)[:, 0]
        np.testing.assert_array_equal(got, np.array(exp, bool), err_msg=f"{w} {red} {op} {vr}")
    tn = np.stack([np.arange(365) + 273.15] * 2, axis=1).astype(np.float32)
    thr = (np.array([330.0, 360.0]) + 273.15).astype(np.float32)
    out = xgen.spell_length_statistics(tn, thr, 1, "min", ">", "sum", TimeAxis.daily("2001-01-01", 365), "YS", device=dev)
    np.testing.assert_allclose(out, [[34, 4]])
    d = _SM_D1[:, None]
    with pytest.raises(ValueError, match="must be a sequence of the same length"):
        xgen.spell_mask([d, d], 3, "min", "<=", 2, device=dev)
    with pytest.raises(ValueError, match="must be a sequence of the same length"):
        xgen.spell_mask([d, d], 3, "min", "<=", [2], device=dev)
    with pytest.raises(ValueError, match="is only supported if 'win_reducer' is 'mean'"):
        xgen.spell_mask(d, 3, "min", "<=", 2, weights=[1, 2, 3], device=dev)
    with pytest.raises(ValueError, match="Weights have a different length"):
        xgen.spell_mask(d, 3, "mean", "<=", 2, weights=[1, 2], device=dev)


@pytest.mark.parametrize("window", [1, 3, 5, 12])
@pytest.mark.parametrize("red,op", [("min", ">"), ("max", "<="), ("mean", ">="), ("sum", "<"), ("max", ">")])
def test_spell_mask_per_cell_thresholds_with_a_window(dev, rng, window, red, op):
    """gen:434-540 with a threshold that has one value per grid cell (in the reference: a DataArray without the time
    dimension, as in tests/test_generic.py:754-766) and a window > 1: rolling statistic, per-cell compare, then "part of
    any window that satisfies the condition".  Masks and the spell statistics built on them equal the oracle."""
    T, shape = 500, (6, 7)
    x = (rng.normal(0, 1, (T,) + shape)).astype(np.float32)
    x[rng.random(x.shape) < 0.01] = np.nan
    x[:, 0, 0] = np.nan
    thr = rng.normal(0.0 if red != "sum" else 0.0 * window, 0.5, shape)          # float64, like a DataArray of thresholds
    got = xgen.spell_mask(x, window, red, op, thr, device=dev)
    exp = ogen.spell_mask(x, window, red, op, thr)
    np.testing.assert_array_equal(got, exp)
    assert 0 < got.mean() < 1
    got_gap = xgen.spell_mask(x, window, red, op, thr, min_gap=3, device=dev)
    np.testing.assert_array_equal(got_gap, ogen.spell_mask(x, window, red, op, thr, min_gap=3))
    ta, ot = TimeAxis.daily("2001-01-01", T), OTime.standard("2001-01-01", T)
    for before in (True, False):
        st = xgen.spell_length_statistics(x, thr, window, red, op, "max", ta, "YS", resample_before_rl=before, device=dev)
        np.testing.assert_array_equal(st, ogen.spell_length_statistics(x, thr, window, red, op, "max", ot, "YS", resample_before_rl=before))
    if window > 1:   # weights (gen:523-524) with per-cell thresholds: the dot product as a field (xh_rolling_dot)
        w = list(rng.random(window) + 0.1)
        np.testing.assert_array_equal(xgen.spell_mask(x, window, "mean", op, thr, weights=w, device=dev),
                                      ogen.spell_mask(x, window, "mean", op, thr, weights=w))


def test_thresholded_events_reference_known_answers(dev):
    """tests/test_generic.py:800-905 (TestThresholdedEvents: simple, different stop window, window_stop 3, freq "MS")
    through the HIP path, and the same numbers from the oracle's find_events on the compare masks."""
    arr = np.array([0, 0, 0, 1, 2, 3, 0, 3, 3, 10, 0, 0, 0, 0, 0, 1, 2, 2, 2, 0, 0, 0, 0, 0, 0, 1, 3, 3, 2, 0, 0, 0, 2, 0, 0, 0, 0],
                   dtype=np.float32)[:, None]

    def check(out, length, eff, total, start):
        keep = ~np.isnan(out["event_length"][..., 0])
        np.testing.assert_array_equal(out["event_length"][..., 0][keep], length)
        np.testing.assert_array_equal(out["event_effective_length"][..., 0][keep], eff)
        np.testing.assert_array_equal(out["event_sum"][..., 0][keep], total)
        np.testing.assert_array_equal(out["event_start"][..., 0][keep], start)

    out = xgen.thresholded_events(arr, 1.0, ">=", 3, device=dev)
    assert out["event_length"].shape[0] == np.ceil(arr.shape[0] / (3 + 1))
    check(out, [3, 3, 4, 4], [3, 3, 4, 4], [6, 16, 7, 9], [3, 7, 15, 25])           # 2000-01-04 / -08 / -16 / -26
    check(xgen.thresholded_events(arr, 2.0, ">=", 3, window_stop=4, device=dev), [3, 3, 7], [3, 3, 4], [16, 6, 10], [7, 16, 26])
    check(xgen.thresholded_events(arr, 1.0, ">=", 3, window_stop=3, device=dev), [7, 4, 4], [6, 4, 4], [22, 7, 9], [3, 15, 25])
    ref = orl.find_events(arr >= 1, 3, arr < 1, 3, data=arr)
    check(ref, [7, 4, 4], [6, 4, 4], [22, 7, 9], [3, 15, 25])
    jan = [0, 0, 0, 1, 2, 3, 0, 3, 3, 10, 0, 0, 0, 0, 0, 0, 2, 2, 2, 2, 2, 2, 0, 0, 0, 0, 0, 3, 2, 3, 2]
    fev = [2, 2, 1, 0, 0, 0, 3, 3, 4, 5, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]
    pr = np.array(jan + fev, dtype=np.float32)[:, None]
    out = xgen.thresholded_events(pr, 1.0, ">=", 3, window_stop=3, freq="MS", time=TimeAxis.daily("2000-01-01", 60), device=dev)
    assert out["event_length"].shape[:2] == (2, 6)
    np.testing.assert_array_equal(out["event_length"][:, :3, 0], [[7, 6, 4], [3, 5, np.nan]])
    np.testing.assert_array_equal(out["event_effective_length"][:, :3, 0], [[6, 6, 4], [3, 5, np.nan]])
    np.testing.assert_array_equal(out["event_sum"][:, :3, 0], [[22, 12, 10], [5, 17, np.nan]])
    np.testing.assert_array_equal(out["event_start"][:, :3, 0], [[3, 16, 27], [0, 6, np.nan]])  # days into the month


@pytest.mark.parametrize("axis", [1, 2, -1])
def test_run_lengths_along_another_axis(dev, rng, axis):
    """rl:223, 275, 338, 381, 437, 643, 693: ``dim`` is any dimension of the DataArray.  The numpy mirrors take the run axis as
    an integer ``dim`` (axis moved first, same kernels, full-shape results get the axis back); the oracle runs along axis 0 of
    the moved array.  Resampling belongs to the time axis and is refused with another ``dim``."""
    from oracle import run_length as orl
    from xclim_amd import run_length as hrl

    m = (rng.random((40, 37, 23)) < 0.6).astype(np.float32)
    m[rng.random(m.shape) < 0.02] = np.nan
    mv = np.moveaxis(m, axis, 0)
    for index in ("first", "last"):
        np.testing.assert_array_equal(hrl.rle(m, axis, index, device=dev), np.moveaxis(orl.rle(mv, index), 0, axis))
        np.testing.assert_array_equal(hrl.rle_statistics(m, "max", 2, axis, None, False, index, device=dev), orl.rle_statistics(mv, "max", 2, index=index, ufunc_1dim=False))
    np.testing.assert_array_equal(hrl.longest_run(m, axis, None, False, device=dev), orl.longest_run(mv, ufunc_1dim=False))
    np.testing.assert_array_equal(hrl.windowed_run_count(m, 3, axis, None, False, device=dev), orl.windowed_run_count(mv, 3, ufunc_1dim=False))
    np.testing.assert_array_equal(hrl.windowed_run_events(m, 3, axis, None, False, device=dev), orl.windowed_run_events(mv, 3, ufunc_1dim=False))
    # (the default dispatch: a grid this small takes the 1-D rules around NaN steps in both, rl:33-78)
    np.testing.assert_array_equal(hrl.longest_run(m, axis, device=dev), orl.longest_run(mv))
    np.testing.assert_array_equal(hrl.first_run(m, 3, axis, device=dev), orl.first_run(mv, 3))
    np.testing.assert_array_equal(hrl.last_run(m, 3, axis, device=dev), orl.last_run(mv, 3))
    np.testing.assert_array_equal(hrl._cumsum_reset(m, axis, "last", device=dev), np.moveaxis(orl.cumsum_reset(mv, "last"), 0, axis))
    from xclim_amd.timeaxis import TimeAxis

    with pytest.raises(ValueError, match="resample the time axis"):
        hrl.longest_run(m, axis, "YS", time=TimeAxis.daily("2001-01-01", m.shape[axis], "noleap"), device=dev)
    with pytest.raises(NotImplementedError, match="integer axis"):
        hrl.longest_run(m, "lat", device=dev)
