"""The reference's own synthetic known-answer tests, restated on the oracle.

Each test names the reference test it ports (paths under /root/reference/tests).  The fixtures mirror
``test_timeseries`` (src/xclim/testing/helpers.py:163-217): daily series starting 2000-07-01 unless stated.
"""

import numpy as np
import pytest

from oracle import calendar as ocal
from oracle import generic as ogen
from oracle import indices as oidx
from oracle import run_length as orl
from oracle.timeutil import OTime

START = "2000-07-01"


def _t(n, start=START):
    return OTime.standard(start, n)


# ---- tests/test_run_length.py ---------------------------------------------------------------------------------
@pytest.mark.parametrize("index", ["first", "last"])
def test_rle(index):  # test_run_length.py:100-130
    values = np.zeros((365, 4, 4))
    values[1:11] = 1
    out = orl.rle(values != 0, index=index).mean(axis=(1, 2))
    expected = np.zeros(365)
    if index == "last":
        expected[1:10] = np.nan
        expected[10] = 10
    else:
        expected[1] = 10
        expected[2:11] = np.nan
    np.testing.assert_array_equal(out, expected)


def test_rle_all_nan():  # test_run_length.py:89-91
    assert (orl.rle(np.full(365, np.nan)) == 0).all()


class TestStatisticsRun:  # test_run_length.py:166-296
    def _both(self, da, freq, reducer, window):
        t = _t(len(da)) if not hasattr(self, "_start") else OTime.standard(self._start, len(da))
        before = orl.resample_and_rl(da, True, orl.rle_statistics, time=t, freq=freq, reducer=reducer, window=window)
        after = orl.rle_statistics(da, reducer, window, time=t, freq=freq)
        return before, after

    def test_simple(self):
        v = np.zeros(365)
        v[1:11] = 1
        for lt in self._both(v != 0, "ME", "max", 1):
            assert lt[0] == 10
            np.testing.assert_array_equal(lt[1:], 0)

    def test_start_at_0(self):
        v = np.zeros(365)
        v[0:10] = 1
        for lt in self._both(v != 0, "ME", "max", 1):
            assert lt[0] == 10
            np.testing.assert_array_equal(lt[1:], 0)

    def test_end_start_at_0(self):
        v = np.zeros(365)
        v[-10:] = 1
        for lt in self._both(v != 0, "ME", "max", 1):
            assert lt[-1] == 10
            np.testing.assert_array_equal(lt[:-1], 0)

    def test_all_true(self):
        v = np.ones(365)
        before, after = self._both(v != 0, "ME", "max", 1)
        np.testing.assert_array_equal(before, [31, 31, 30, 31, 30, 31, 31, 28, 31, 30, 31, 30])
        expected = np.zeros(12)
        expected[0] = 365
        np.testing.assert_array_equal(after, expected)

    def test_almost_all_true(self):
        v = np.ones(365)
        v[35] = 0
        before, after = self._both(v != 0, "ME", "max", 1)
        assert before[0] == 31 and before[1] == 26
        assert after[0] == 35 and after[1] == 365 - 35 - 1

    def test_other_stats(self):
        v = np.ones(365)
        v[35] = 0
        self._start = "2000-01-01"
        try:
            for lt in self._both(v != 0, "YS", "min", 1):
                assert lt == 35
            for lt in self._both(v != 0, "YS", "mean", 36):
                assert lt == 329
            for lt in self._both(v != 0, "YS", "std", 1):
                assert lt == 147
            _, q90 = self._both(v != 0, "YS", "q90", 1)
            _, q10 = self._both(v != 0, "YS", "q10", 1)
            assert q90 == 299.6 and q10 == 64.4
        finally:
            del self._start

    @pytest.mark.parametrize("op", ["min", "max"])
    def test_resampling_order(self, op):
        self._start = "2000-01-01"
        try:
            v = np.ones(365)
            v[35:45] = 0
            b, a = self._both(v != 0, "MS", op, 1)
            assert (b != a).any()
            v = np.zeros(365)
            v[0:-1:31] = 1
            b, a = self._both(v != 0, "MS", op, 1)
            assert (b == a).any()
        finally:
            del self._start


def test_first_run():  # test_run_length.py:299-353
    a = np.zeros(100, bool)
    a[10:20] = 1
    assert orl.first_run(a, 5) == 10
    t = np.zeros(60)
    t[30:40] = 2
    runs = np.stack([t, t], axis=1)
    np.testing.assert_array_equal(orl.first_run(runs, 1), [30, 30])
    t[0] = 2
    runs = np.stack([t, t], axis=1)
    out = orl.first_run(runs, 1, time=OTime.standard("2000-01-01", 60), freq="MS")
    np.testing.assert_array_equal(out, [[0, 0], [0, 0]])


def test_last_run():  # test_run_length.py:384-424
    t = np.zeros(60)
    t[30:40] = 2
    runs = np.stack([t, t], axis=1)
    np.testing.assert_array_equal(orl.last_run(runs, 1), [39, 39])
    t[0] = 2
    runs = np.stack([t, t], axis=1)
    out = orl.last_run(runs, 1, time=OTime.standard("2000-01-01", 60), freq="MS")
    np.testing.assert_array_equal(out, [[30, 30], [8, 8]])


@pytest.mark.parametrize("index", ["first", "last"])
def test_windowed_run_events_count_sum(index):  # test_run_length.py:356-381
    a = np.zeros(50, bool)
    a[4:7] = True
    a[34:45] = True
    assert orl.windowed_run_events(a, 3, index=index) == 2
    assert orl.windowed_run_count(a, 3, index=index) == 3 + 11
    f = np.zeros(50, float)
    f[4:6] = 5
    f[25:30] = 5
    f[35:45] = 5
    assert orl.windowed_max_run_sum(f, 3, index=index) == 50


# ---- tests/test_generic.py ----------------------------------------------------------------------------------------
def test_threshold_count_and_domain_count():  # test_generic.py:70-82
    ts = np.arange(365).astype(np.float64)
    np.testing.assert_array_equal(ogen.threshold_count(ts, "<", 50, _t(365), "YE"), [50, 0])
    np.testing.assert_array_equal(ogen.domain_count(ts, 10, 20, _t(365), "YE"), [10, 0])


def test_get_op_errors():  # generic.py:285, 296
    with pytest.raises(ValueError, match="not recognized"):
        ogen.get_op("=>")
    with pytest.raises(ValueError, match="not permitted"):
        ogen.threshold_count(np.arange(3.0), "==", 1, _t(3), "YS")


# ---- tests/test_calendar.py -----------------------------------------------------------------------------------
def test_percentile_doy():  # test_calendar.py:83-104
    tas = np.arange(365).astype(np.float64)
    tas2 = np.stack([tas, tas], axis=1)
    ot = OTime.standard("2001-01-01", 365)
    p, doys = ocal.percentile_doy(tas2, ot, window=5, per=50)
    assert p[list(doys).index(3), 0, 0] == 2
    tasn = tas.copy()
    tasn[ot.doy == 2] = np.nan
    p, doys = ocal.percentile_doy(np.stack([tasn, tasn], axis=1), ot, window=5, per=50)
    assert p[list(doys).index(3), 0, 0] == 2.5


# ---- tests/test_indices.py ------------------------------------------------------------------------------------
class TestMaximumConsecutiveDryDays:  # test_indices.py:2354-2381
    thr = 1.0 / 86400.0  # "1 mm/day" in kg m-2 s-1 (hydro context)

    def test_simple(self):
        a = np.zeros(365) + 10
        a[5:15] = 0
        out = oidx.maximum_consecutive_dry_days(a.astype(np.float32), self.thr, _t(365), "ME")
        assert out[0] == 10

    def test_run_start_at_0(self):
        a = np.zeros(365) + 10
        a[:10] = 0
        assert oidx.maximum_consecutive_dry_days(a.astype(np.float32), self.thr, _t(365), "ME")[0] == 10

    @pytest.mark.parametrize("before,expected", [(True, 26), (False, 30)])
    def test_resampling_order(self, before, expected):
        a = np.zeros(365) + 10
        a[5:35] = 0
        out = oidx.maximum_consecutive_dry_days(a.astype(np.float32), self.thr, _t(365), "ME", resample_before_rl=before)
        assert out[0] == expected


class TestTGXNp:  # test_indices.py:2529-2625 (leap year: 366 -> 365 + re-interpolation path of percentile_doy)
    def _setup(self):
        tas = np.arange(366).astype(np.float64)
        ot = OTime.standard("2000-01-01", 366)
        p, doys = ocal.percentile_doy(tas, ot, per=10)
        assert len(doys) == 366
        tas = tas.copy()
        tas[175:180] = 1  # cold spell in June
        return tas, p[..., 0], doys, ot

    def test_t10p(self):
        tas, t10, doys, ot = self._setup()
        out = oidx.tx10p(tas, t10, doys, ot, "MS", op="<")
        assert out[0] == 0 and out[5] == 5

    def test_t90p(self):
        tas, t10, doys, ot = self._setup()
        out = oidx.tx90p(tas, t10, doys, ot, "MS", op=">")
        assert out[0] == 30 and out[1] == 29 and out[5] == 25


def test_tg_mean_and_missing():  # test_temperature.py:162-191 semantics: NaN day -> period masked (MissingAny)
    tas = np.full(731, 280.0, np.float32)
    tas[400] = np.nan
    ot = OTime.standard("2001-01-01", 731)
    out = oidx.apply_missing(oidx.tg_mean(tas, ot, "YS"), tas, ot, "YS")
    assert out[0] == 280.0 and np.isnan(out[1]) and np.isnan(out[2])  # 2002 has the NaN, 2003 is a 1-day stub


def test_range_reductions_known_answers():
    """Hand-computed answers for the two-variable range reductions (gen:1076-1105, 1360-1414), in the style of
    tests/test_indices.py::TestDailyTemperatureRange / TestETR / TestVDTR of the reference."""
    from oracle import generic as ogen
    from oracle.timeutil import OTime

    T = 10
    low = np.full((T, 1), 280.0, np.float32)
    high = low + np.arange(T, dtype=np.float32)[:, None]  # ranges 0..9
    high[3, 0] = np.nan
    t = OTime.standard("2000-01-01", T)
    assert ogen.diurnal_temperature_range(low, high, "max", t, "YS")[0, 0] == 9
    assert ogen.diurnal_temperature_range(low, high, "mean", t, "YS")[0, 0] == np.float32((45 - 3) / 9)
    # |diff| of (0,1,2,nan,4,...,9): 1,1,nan,nan,1,1,1,1,1 -> mean 1
    assert ogen.interday_diurnal_temperature_range(low, high, t, "YS")[0, 0] == 1
    assert ogen.extreme_temperature_range(low, high, t, "YS")[0, 0] == 9
    ev = ogen.get_daily_events(high[:, 0], 283.5, ">")
    assert np.isnan(ev[3]) and ev[:3].sum() == 0 and ev[4:].sum() == 6


def test_sdba_oracle_against_numpy_and_the_reference_testqm():
    """The EQM oracle is a SPECIFIED restatement (xsdba is not in the tree).  What can be pinned without it:
    (1) nbutils.quantile is documented as equivalent to np.nanquantile (linear / Hyndman-Fan type 7): the restatement
        must agree with numpy's own implementation;
    (2) the reference's only numeric test of the path (tests/test_xsdba.py:113-155, TestQM.test_quantiles: train on
        hist ~ U(10, 11), ref ~ N(12, 1), 50 quantiles, adjust with interp="linear", compare to 1 decimal in the interior),
        restated on the oracle for both kinds."""
    from scipy.stats import norm, uniform

    from oracle import sdba as osdba

    rng = np.random.default_rng(42)
    x = rng.normal(0, 1, (500, 7)).astype(np.float32)
    x[rng.random(x.shape) < 0.1] = np.nan
    x[:, 3] = np.nan
    q = osdba.equally_spaced_nodes(20)
    with np.errstate(all="ignore"):
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            ref_np = np.nanquantile(x.astype(np.float64), q, axis=0)
    np.testing.assert_allclose(osdba.quantile(x, q), ref_np, rtol=1e-6, equal_nan=True)
    np.testing.assert_allclose(q, (np.arange(20) + 0.5) / 20)

    u = rng.random(10000)
    xs = uniform.ppf(u, loc=10, scale=1)
    ys = norm.ppf(u, loc=12, scale=1)
    for kind in ("+", "*"):
        af, hq = osdba.eqm_train(ys[:, None].astype(np.float32), xs[:, None].astype(np.float32), 50, kind)
        qq = osdba.equally_spaced_nodes(50)
        expected = (norm.ppf(qq, 12, 1) - uniform.ppf(qq, 10, 1)) if kind == "+" else (norm.ppf(qq, 12, 1) / uniform.ppf(qq, 10, 1))
        np.testing.assert_array_almost_equal(af[2:-2, 0], expected[2:-2], 1)
        p = osdba.eqm_adjust(xs[:, None].astype(np.float32), af, hq, kind, "linear", "constant")
        middle = (u > 1e-2) & (u < 0.99)
        np.testing.assert_array_almost_equal(p[middle, 0], ys[middle], 1)


def test_wet_percentile_and_heat_wave_indices_known_answers():
    """SURVEY 8f rank 1, pinned by the reference's own known answers:
    tests/test_indices.py:1579-1614 (TestDaysOverPrecipThresh) and :1859-1960 (TestHeatWave{Frequency,MaxLength,TotalLength})."""
    from oracle import indices as oidx
    from oracle.timeutil import OTime

    K2C = 273.15
    a = np.zeros(365, dtype=np.float32)
    a[:8] = np.arange(8)
    t = OTime.standard("2000-01-01", 365)
    per = np.zeros(366)
    per[5:] = 5
    doys = np.arange(1, 367)
    assert oidx.days_over_precip_thresh(a, per, doys, t, 2.0)[0] == 4
    np.testing.assert_array_almost_equal(oidx.fraction_over_precip_thresh(a, per, doys, t, 2.0)[0], (3 + 4 + 6 + 7) / (3 + 4 + 5 + 6 + 7))
    # test_quantile: a per-cell percentile of 5 -> only days 6 and 7 qualify
    assert oidx.days_over_precip_thresh(a, np.float64(5.0), None, t, 2.0)[0] == 2
    # test_nd: pr = 1 everywhere, percentile 0 -> floored by the 0.5 wet-day threshold, all 300 days count
    t300 = OTime.standard("2000-01-01", 300)
    out = oidx.days_over_precip_thresh(np.ones((300, 2, 3), np.float32), np.zeros((2, 3)), None, t300, 0.5)
    np.testing.assert_array_equal(out, np.full((1, 2, 3), 300))

    tn = (np.asarray([20, 23, 23, 23, 23, 22, 23, 23, 23, 23]) + K2C).astype(np.float32)
    tx = (np.asarray([29, 31, 31, 31, 29, 31, 31, 31, 31, 31]) + K2C).astype(np.float32)
    t10 = OTime.standard("2000-01-01", 10)
    for (a_, b_, w), f, m, tot in [((22, 30, 3), 2, 4, 7), ((10, 10, 3), 1, 10, 10), ((40, 40, 3), 0, 0, 0)]:
        assert oidx.heat_wave_frequency(tn, tx, t10, a_ + K2C, b_ + K2C, w)[0] == f
        assert oidx.heat_wave_max_length(tn, tx, t10, a_ + K2C, b_ + K2C, w)[0] == m
        assert oidx.heat_wave_total_length(tn, tx, t10, a_ + K2C, b_ + K2C, w)[0] == tot
    assert oidx.heat_wave_frequency(tn, tx, t10, 22 + K2C, 30 + K2C, 4)[0] == 1
    assert oidx.heat_wave_max_length(tn, tx, t10, 22 + K2C, 30 + K2C, 5)[0] == 0
    assert oidx.heat_wave_total_length(tn, tx, t10, 22 + K2C, 30 + K2C, 5)[0] == 0


def test_hot_spell_max_magnitude_known_answer():
    """tests/test_indices.py:2132-2142 (TestHotSpellMaxMagnitude; the series fixture starts on 2000-07-01, monthly periods)."""
    from oracle import indices as oidx
    from oracle.timeutil import OTime

    a = np.zeros(365)
    a[15:20] += 30
    a[40:42] += 50
    a[86:96] += 30
    da = (a + 273.15).astype(np.float32)
    out = oidx.hot_spell_max_magnitude(da, 25 + 273.15, OTime.standard("2000-07-01", 365), 3, "MS")
    np.testing.assert_allclose(out, [25, 0, 30, 20, 0, 0, 0, 0, 0, 0, 0, 0], atol=1e-3)


def test_select_time_known_answers():
    """core/calendar.py:1259-1378 pinned by the reference's tests/test_generic.py:523-690 (month, season, doy bounds that
    wrap over the year end, date bounds with Feb 29 on a standard calendar) — for the oracle restatement AND the host
    mirror's mask builder (pure calendar arithmetic, no device)."""
    import pandas as pd

    from oracle import calendar as ocal
    from oracle.timeutil import OTime
    from xclim_amd.calendar import select_time_mask
    from xclim_amd.timeaxis import TimeAxis

    def both(start, end, **indexer):
        idx = pd.date_range(start, end)
        ot, ta = OTime.standard(start, len(idx)), TimeAxis.daily(start, len(idx))
        m1, m2 = ocal.select_time_mask(ot, **indexer), select_time_mask(ta, **indexer)
        np.testing.assert_array_equal(m1, m2)
        return idx[m1]

    def span(*pairs):
        return pd.DatetimeIndex(np.concatenate([pd.date_range(a, b).values for a, b in pairs]))

    sel = both("1993-01-05", "1994-12-31", month=1)
    assert len(sel) == 58 and sel.equals(span(("1993-01-05", "1993-01-31"), ("1994-01-01", "1994-01-31")))
    sel = both("1993-01-05", "1994-12-31", season="DJF")
    assert sel.equals(span(("1993-01-05", "1993-02-28"), ("1993-12-01", "1994-02-28"), ("1994-12-01", "1994-12-31")))
    sel = both("2003-02-13", "2004-12-31", doy_bounds=(360, 75))
    assert sel.equals(span(("2003-02-13", "2003-03-16"), ("2003-12-26", "2004-03-15"), ("2004-12-25", "2004-12-31")))
    sel = both("2003-02-13", "2004-12-31", doy_bounds=(25, 80))
    assert sel.equals(span(("2003-02-13", "2003-03-21"), ("2004-01-25", "2004-03-20")))
    sel = both("2003-02-13", "2005-11-01", date_bounds=("10-05", "02-29"))
    assert sel.equals(span(("2003-02-13", "2003-02-28"), ("2003-10-05", "2004-02-29"), ("2004-10-05", "2005-02-28"),
                           ("2005-10-05", "2005-11-01")))
    sel = both("1990-01-01", "1993-12-31", date_bounds=("02-29", "03-02"))  # the docstring example (cal:1316-1323)
    assert [str(d.date()) for d in sel] == ["1990-03-01", "1990-03-02", "1991-03-01", "1991-03-02", "1992-02-29", "1992-03-01",
                                            "1992-03-02", "1993-03-01", "1993-03-02"]
    # noleap / 360_day axes: seasons and months by month number
    ot, ta = OTime.noleap(1993, 730), TimeAxis.daily("1993-01-01", 730, "noleap")
    m = ocal.select_time_mask(ot, season=["MAM", "SON"])
    np.testing.assert_array_equal(m, select_time_mask(ta, season=["MAM", "SON"]))
    assert m.sum() == 2 * (92 + 91)
    ot, ta = OTime.noleap(1993, 720, "360_day"), TimeAxis.daily("1993-01-01", 720, "360_day")
    m = ocal.select_time_mask(ot, month=[3, 6])
    np.testing.assert_array_equal(m, select_time_mask(ta, month=[3, 6]))
    assert m.sum() == 120
    with pytest.raises(ValueError, match="Only one method"):
        select_time_mask(ta, month=1, season="DJF")
    assert select_time_mask(ta) is None


_DRY_SPELL_CASES = [
    ([1.01] * 6 + [0.01] * 3 + [0.51] * 2 + [0.75] * 2 + [0.51] + [0.01] * 3 + [1.01] * 3, 3, 3, 7, (1, 12, 20, 12, 20)),
    ([0.01] * 6 + [1.01] * 3 + [0.51] * 2 + [0.75] * 2 + [0.51] + [0.01] * 3 + [0.01] * 3, 3, 3, 7, (2, 18, 20, 10, 20)),
    ([3.01] * 358 + [0.99] * 14 + [3.01] * 358, 1, 14, 14, (0, 7, 7, 7, 7)),
]
_DRY_FREQ_OP = [29.012, 0.1288, 0.0253, 0.0035, 4.9147, 1.4186, 1.014, 0.5622, 0.8001, 10.5823, 2.8879, 8.2635, 0.292, 0.5242,
                0.2426, 1.3934, 0.0, 0.4633, 0.1862, 0.0034, 2.4591, 3.8547, 3.1983, 3.0442, 7.422, 14.8854, 13.4334, 0.0012,
                0.0782, 31.2916, 0.0379]


def test_dry_spell_known_answers():
    """tests/test_indices.py:4067-4171 (test_dry_spell, the indexer variants, test_dry_spell_frequency_op) on the oracle's
    spell_length_statistics: window sums / maxima under a threshold, spell count / total length / longest spell."""
    from oracle import generic as ogen
    from oracle.timeutil import OTime

    for pr, th1, th2, window, outs in _DRY_SPELL_CASES:
        x = np.asarray(pr, dtype=np.float32)
        ot = OTime.standard("1981-01-01", len(x))
        f = lambda th, wop, red: ogen.spell_length_statistics(x, np.float32(th), window, wop, "<", red, ot, "YS")[0]
        got = (f(th1, "sum", "count"), f(th2, "sum", "sum"), f(th1, "max", "sum"), f(th2, "sum", "max"), f(th1, "max", "max"))
        np.testing.assert_allclose(got, outs, rtol=1e-1)
    x = np.asarray([1] * 5 + [0] * 10 + [1] * 350, dtype=np.float32)
    ot = OTime.standard("1900-01-01", len(x))
    for red in ("sum", "max"):
        out = ogen.spell_length_statistics(x, np.float32(3.1), 7, "sum", "<", red, ot, "MS", date_bounds=("01-10", "12-31"))
        np.testing.assert_allclose(out, [9] + [0] * 11)
    x = np.asarray(_DRY_FREQ_OP, dtype=np.float32)
    ot = OTime.standard("2000-07-01", len(x))
    assert ogen.spell_length_statistics(x, np.float32(1.0), 3, "sum", "<", "count", ot, "MS")[0] == 2
    assert ogen.spell_length_statistics(x, np.float32(1.0), 3, "max", "<", "count", ot, "MS")[0] == 3


K2C = 273.15
_HS_FREQ = ([29, 31, 31, 31, 29, 31, 31, 31, 31, 31],
            [(30, 3, ">", 2), (30, 4, ">", 1), (29, 3, ">", 2), (29, 3, ">=", 1), (10, 3, ">", 1), (40, 5, ">", 0)])
_HS_LEN = ([28, 31, 31, 31, 29, 31, 31, 31, 31, 31],
           [(30, 3, ">", 5, 8), (10, 3, ">", 10, 10), (29, 3, ">", 5, 8), (29, 3, ">=", 9, 9), (40, 3, ">", 0, 0), (30, 5, ">", 5, 5)])


def _spell_series():
    hot = np.zeros(365)
    hot[10:20] += 30
    hot[40:43] += 50
    hot[80:100] += 30
    cold_f = np.zeros(365)
    cold_f[10:20] -= 15
    cold_f[40:43] -= 50
    cold_f[80:86] -= 30
    cold_f[95:101] -= 30
    order = np.zeros(365)
    order[5:35] = 31
    return (hot + K2C).astype(np.float32), (-hot * 0 + K2C).astype(np.float32), (cold_f + K2C).astype(np.float32), (order + K2C).astype(np.float32)


def test_hot_and_cold_spell_known_answers():
    """tests/test_indices.py:119-147 (cold_spell_days / cold_spell_frequency) and :2040-2131 (hot_spell_frequency /
    max_length / total_length, incl. the resampling-order case) on the oracle compositions."""
    from oracle import indices as oidx
    from oracle.timeutil import OTime

    t10 = OTime.standard("2000-07-01", 10)
    tx = (np.asarray(_HS_FREQ[0]) + K2C).astype(np.float32)
    for th, w, op, exp in _HS_FREQ[1]:
        assert oidx.run_index(tx, op, np.float32(th + K2C), "events", w, t10, "YS")[0] == exp
    tx = (np.asarray(_HS_LEN[0]) + K2C).astype(np.float32)
    for th, w, op, mx, tot in _HS_LEN[1]:
        assert oidx.longest_run_index(tx, op, np.float32(th + K2C), w, t10, "YS")[0] == mx
        assert oidx.run_index(tx, op, np.float32(th + K2C), "count", w, t10, "YS")[0] == tot
    hot, _, cold_f, order = _spell_series()
    t = OTime.standard("2000-07-01", 365)
    np.testing.assert_array_equal(oidx.run_index(hot, ">", np.float32(25 + K2C), "count", 5, t, "MS"), [10, 0, 12, 8] + [0] * 8)
    cold = (2 * K2C - hot.astype(np.float64)).astype(np.float32)  # the same pattern below zero: -30 / -50 C
    np.testing.assert_array_equal(oidx.run_index(cold, "<", np.float32(-10 + K2C), "count", 5, t, "MS"), [10, 0, 12, 8] + [0] * 8)
    t71 = OTime.standard("1971-01-01", 365)
    np.testing.assert_array_equal(oidx.run_index(cold_f, "<", np.float32(-10 + K2C), "events", 5, t71, "MS"), [1, 0, 1, 1] + [0] * 8)
    assert oidx.run_index(cold_f, "<", np.float32(-10 + K2C), "events", 5, t71, "YS")[0] == 3
    assert oidx.run_index(order, ">", np.float32(30 + K2C), "events", 3, t, "MS", True)[1] == 1
    assert oidx.run_index(order, ">", np.float32(30 + K2C), "events", 3, t, "MS", False)[1] == 0


_GSL_CASES = [("1950-01-01", "1951-01-01", 0), ("2000-01-01", "2000-12-31", 365), ("2000-07-10", "2001-01-01", 0),
              ("2000-06-15", "2001-01-01", 199), ("2000-06-15", "2000-07-15", 31)]


def _gsl_series(d1, d2, T=365, start="2000-01-01", cold=0.0, warm=280.0):
    import pandas as pd

    idx = pd.date_range(start, periods=T)
    return np.where((idx >= d1) & (idx <= d2), warm, cold).astype(np.float32)


def test_growing_season_length_known_answers():
    """tests/test_indices.py:1681-1707 (growing_season_length: thresh 5 degC, window 6, mid_date 07-01) on the oracle's
    season restatement, incl. the southern-hemisphere case (mid_date 01-01, freq YS-JUL -> 121)."""
    from oracle import run_length as orl
    from oracle.timeutil import OTime

    ot = OTime.standard("2000-01-01", 365)
    for d1, d2, exp in _GSL_CASES:
        tas = _gsl_series(d1, d2)
        assert orl.season(tas >= np.float32(278.15), 6, "07-01", ot)[2] == exp
    tas = _gsl_series("2000-11-01", "2001-03-01", T=730)
    ot2 = OTime.standard("2000-01-01", 730)
    length = orl.season_per_period(tas >= np.float32(278.15), 6, "01-01", ot2, "YS-JUL")[2]
    assert length[1] == 121  # the period starting 2000-07-01


def _rwh_series():
    values = np.zeros(365)
    a = [0, 1, 0, 1, 1, 0, 0, 1, 1, 1, 1, 0, 0, 0, 0, 1, 1, 1, 1, 1, 0, 0, 0, 0]
    values[: len(a)] = a
    expected = values * 0
    expected[1:11] = 1
    expected[15:20] = 1
    return values, expected


_SEASON_LENGTH = [("07-01", 210, 70), ("07-01", 190, 50), ("04-01", 150, 0), ("11-01", 150, 165), (None, 150, 10)]
_RUN_END_AFTER = [("07-01", 210, 210), ("07-01", 190, 190), ("04-01", 150, np.nan), ("11-01", 150, 305)]   # indices (doy - 1)
_FIRST_RUN_AFTER = [("07-01", 210, 210), ("07-01", 190, 190), ("04-01", False, np.nan), ("11-01", 150, 305)]


def test_more_run_length_known_answers():
    """tests/test_run_length.py:135-160 (runs_with_holes), :427-434 (run_bounds), :451-454 (keep_longest_run), :472-560
    (season_length / run_end_after_date / first_run_after_date with dates) on the oracle."""
    from oracle import run_length as orl
    from oracle.timeutil import OTime

    values, expected = _rwh_series()
    np.testing.assert_array_equal(orl.runs_with_holes(values == 1, 1, values == 0, 3), expected)
    ident = np.zeros((365, 4))
    ident[1:11] = 1
    np.testing.assert_array_equal(orl.runs_with_holes(ident != 0, 1, ident == 0, 1), ident)
    runs = np.array([0, 1, 1, 1, 0, 0, 1, 1, 1, 0], dtype=bool)
    np.testing.assert_array_equal(orl.keep_longest_run(runs), [0, 1, 1, 1, 0, 0, 0, 0, 0, 0])
    np.testing.assert_array_equal(orl.run_bounds(runs), [[1, 6], [4, 9]])
    ot = OTime.standard("2000-01-01", 360)
    for date, end, exp in _SEASON_LENGTH:
        t = np.zeros(360)
        t[140:end] = 1
        assert orl.season(t == 1, 1, date, ot)[2] == exp, (date, end)
    for date, end, exp in _RUN_END_AFTER:
        t = np.zeros(360)
        t[140:end] = 1
        np.testing.assert_array_equal(orl.run_end_after_date(t == 1, 1, date, ot), exp)
    ot5 = OTime.standard("2000-01-01", 365)
    for date, beg, exp in _FIRST_RUN_AFTER:
        t = np.zeros(365)
        if beg:
            t[beg:] = 1
        np.testing.assert_array_equal(orl.first_run_after_date(t == 1, 1, date, ot5), exp)


_FIND_EVENTS_COND = np.array([[0, 0, 0, 1, 1, 1, 1, 0, 0, 0, 0],   # normal
                              [0, 0, 1, 1, 0, 0, 1, 1, 1, 1, 0],   # two events, one short, one long
                              [0, 1, 1, 1, 1, 0, 0, 1, 0, 0, 0]]).T == 1   # (time, cell)


def test_season_and_find_events_known_answers():
    """tests/test_run_length.py:675-690 (rl.season: start 140, end 150, length 10) and :694-730 (find_events: event
    lengths / starts for window 1 and for window 2 with window_stop 3) on the oracle."""
    from oracle import run_length as orl
    from oracle.timeutil import OTime

    t = np.zeros(360)
    t[140:150] = 1
    beg, end, length = orl.season(t >= 1, 2, None, OTime.standard("2000-01-01", 360))
    assert (beg, end, length) == (140, 150, 10)
    ev = orl.find_events(_FIND_EVENTS_COND, 1)
    exp = np.pad(np.array([[4, np.nan], [2, 4], [4, 1]]), [(0, 0), (0, 4)], constant_values=np.nan).T
    np.testing.assert_equal(ev["event_length"], exp)
    np.testing.assert_equal(ev["event_start"][0], [3, 2, 1])
    ev = orl.find_events(_FIND_EVENTS_COND, 2, None, 3)
    np.testing.assert_equal(ev["event_length"], np.pad(np.array([[4.0], [9.0], [7.0]]), [(0, 0), (0, 2)], constant_values=np.nan).T)


_SM_D1 = np.array([0, 1, 2, 3, 2, 1, 0, 0], dtype=np.float32)
_SM_D2 = np.array([1, 2, 3, 2, 1, 0, 0, 0], dtype=np.float32)
_SM_SINGLE = [((3, "min", ">=", 2, None), [0, 0, 1, 1, 1, 0, 0, 0]), ((3, "max", ">=", 2, None), [1, 1, 1, 1, 1, 1, 1, 0]),
              ((2, "mean", ">=", 2, None), [0, 0, 1, 1, 1, 0, 0, 0]), ((3, "mean", ">", 2, [0.2, 0.4, 0.4]), [0, 1, 1, 1, 1, 0, 0, 0])]
_SM_MULTI = [((3, "min", ">=", [2, 2], None, "all"), [0] * 8), ((3, "min", ">=", [2, 2], None, "any"), [0, 1, 1, 1, 1, 0, 0, 0]),
             ((2, "mean", ">=", [2, 2], None, "all"), [0, 0, 1, 1, 0, 0, 0, 0]),
             ((3, "mean", ">", [2, 1.5], [0.2, 0.4, 0.4], "all"), [0, 1, 1, 1, 1, 0, 0, 0])]


def test_spell_mask_known_answers():
    """tests/test_generic.py:702-732 (TestSpellMask, one and two variables, weights, var_reducer) and :754-766
    (spell_length_statistics with one threshold per site: [34, 4]) on the oracle."""
    from oracle import generic as ogen
    from oracle.timeutil import OTime

    for (w, red, op, thr, weights), exp in _SM_SINGLE:
        np.testing.assert_array_equal(ogen.spell_mask(_SM_D1, w, red, op, np.float32(thr), weights=weights), np.array(exp, bool))
    for (w, red, op, thr, weights, vr), exp in _SM_MULTI:
        got = ogen.spell_mask([_SM_D1, _SM_D2], w, red, op, [np.float32(t) for t in thr], weights=weights, var_reducer=vr)
        np.testing.assert_array_equal(got, np.array(exp, bool))
    tn = np.stack([np.arange(365) + 273.15] * 2, axis=1).astype(np.float32)
    thr = (np.array([330.0, 360.0]) + 273.15).astype(np.float32)
    out = ogen.spell_length_statistics(tn, thr, 1, "min", ">", "sum", OTime.standard("2001-01-01", 365), "YS")
    np.testing.assert_allclose(out, [[34, 4]])


# ---- tests/test_missing.py: the other missing-value methods (core/missing.py:325-512) -----------------------------
def test_missing_wmo_answers():  # test_missing.py:166-197
    from oracle import missing as omiss

    a = np.arange(360.0)
    a[5:7] = np.nan      # under the limit
    a[40:45] = np.nan    # too many consecutive missing values
    a[70:92:2] = np.nan  # too many non-consecutive missing values
    out = omiss.missing_wmo(a, _t(360), "MS")
    assert not out[0] and out[1] and out[2]
    a = np.arange(350.0)
    a[5:16] = np.nan
    np.testing.assert_array_equal(omiss.missing_wmo(a, _t(350), "QS-JAN"), [True, False, False, True])
    np.testing.assert_array_equal(omiss.missing_wmo(np.arange(31.0), _t(31), "YS"), [True])   # one complete month of a year


def test_missing_pct_answers():  # test_missing.py:199-221
    from oracle import missing as omiss

    a = np.arange(360.0)
    a[5:7] = np.nan
    a[40:45] = np.nan
    out = omiss.missing_pct(a, _t(360), "MS", tolerance=0.1)
    assert not out[0] and out[1]
    # a series without the months 5 .. 11 (test_missing_period): months inside the span are expected in full
    full = OTime.standard("2000-01-01", 366)
    keep = np.isin(full.month, [1, 2, 3, 4, 12])
    np.testing.assert_array_equal(omiss.missing_pct(np.ones(keep.sum()), full.isel(keep), "MS", tolerance=0.9), [False] * 4 + [True] * 7 + [False])


def test_at_least_n_valid_and_some_but_not_all_answers():  # test_missing.py:224-231, 274-285
    from oracle import missing as omiss

    a = np.arange(360.0)
    a[5:10] = np.nan
    a[40:55] = np.nan
    np.testing.assert_array_equal(omiss.at_least_n_valid(a, _t(360), "MS", n=20)[:2], [False, True])
    a = np.arange(360.0)
    a[:40] = np.nan
    out = omiss.missing_some_but_not_all(a, _t(360), "MS")
    assert not out[0] and out[1] and not out[2]    # all missing: ok; some missing: flagged; none missing: ok


def test_missing_any_answers_with_indexers_and_without_freq():  # test_missing.py:56-145
    from oracle import missing as omiss

    a = np.arange(360.0)
    a[5:10] = np.nan
    out = omiss.missing_any(a, _t(360), "MS")
    assert out[0] and not out[1]
    np.testing.assert_array_equal(omiss.missing_any(np.arange(66.0), OTime.standard("2001-12-30", 66), "MS"), [True, False, False, True])
    np.testing.assert_array_equal(omiss.missing_any(np.arange(378.0), OTime.standard("2001-12-31", 378), "YS"), [True, False, True])
    z = np.zeros(36)
    np.testing.assert_array_equal(omiss.missing_any(z, _t(36), "YS", month=7), [False])
    np.testing.assert_array_equal(omiss.missing_any(z, _t(36), "YS", month=8), [True])
    np.testing.assert_array_equal(omiss.missing_any(z, _t(36), "YS", month=[7, 8]), [True])
    np.testing.assert_array_equal(omiss.missing_any(np.zeros(76), _t(76), "YS", month=[7, 8]), [False])
    # (the reference converts 2000-01-01 .. 2000-12-25 by date: 359 days without leap day, 355 in the 360-day calendar)
    for ot in (OTime.standard("2000-01-01", 360), OTime.noleap(2000, 359), OTime.noleap(2000, 355, "360_day")):
        np.testing.assert_array_equal(omiss.missing_any(np.zeros(len(ot)), ot, "YS", season="MAM"), [False])
        np.testing.assert_array_equal(omiss.missing_any(np.zeros(len(ot)), ot, "YS", season="DJF"), [True])
    np.testing.assert_array_equal(omiss.missing_any(np.zeros(360), _t(360), None), [False])
    t = list(range(31))
    t.pop(5)
    np.testing.assert_array_equal(omiss.missing_any(np.zeros(30), _t(360).isel(t), None), [True])
    np.testing.assert_array_equal(omiss.missing_any(np.zeros(360), _t(360), None, month=[7]), [False])
    np.testing.assert_array_equal(omiss.missing_any(np.zeros(30), _t(360).isel(t), None, month=[7]), [True])
