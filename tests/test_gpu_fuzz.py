"""Seeded differential fuzzing of the HIP path against the oracle: random shapes (incl. tiny and ragged), NaN fractions,
windows, operators, frequencies and calendars for the run-length, spell, count and percentile families.  Every case is
checked with the family's parity bar (bit-exact integers, <= 1e-6 relative for floats)."""
import numpy as np
import pytest

from oracle import calendar as ocal
from oracle import generic as ogen
from oracle import run_length as orl
from oracle.timeutil import OTime
from xclim_amd import generic as xgen
from xclim_amd import run_length as xrl
from xclim_amd.calendar import percentile_doy
from xclim_amd.timeaxis import TimeAxis

pytestmark = pytest.mark.gpu

OPS = [">", "<", ">=", "<=", "==", "!="]
FREQS = ["YS", "MS", "QS-DEC", "YS-JUL"]


def _case(rng):
    T = int(rng.choice([1, 2, 7, 31, 59, 200, 365, 366, 730, 800]))
    cells = tuple(int(v) for v in rng.choice([1, 2, 3, 4, 5, 8, 13, 64], size=int(rng.integers(1, 3))))
    nanf = float(rng.choice([0.0, 0.0, 0.02, 0.3]))
    start = str(rng.choice(["2000-01-01", "2001-03-15", "1999-12-31", "2004-02-28"]))
    return T, cells, nanf, start


def _axes(start, T):
    return TimeAxis.daily(start, T, "standard"), OTime.standard(start, T)


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_counts_and_reductions(dev, seed):
    rng = np.random.default_rng(1000 + seed)
    for _ in range(6):
        T, cells, nanf, start = _case(rng)
        ta, ot = _axes(start, T)
        x = rng.integers(-3, 4, (T,) + cells).astype(np.float32) + rng.choice([0.0, 0.25]).astype(np.float32)
        x[rng.random(x.shape) < nanf] = np.nan
        freq = str(rng.choice(FREQS))
        op = str(rng.choice(OPS[:4]))
        thr = float(rng.choice([-1.0, 0.0, 0.25, 1.0]))
        np.testing.assert_array_equal(xgen.threshold_count(x, op, thr, ta, freq, device=dev), ogen.threshold_count(x, op, thr, ot, freq),
                                      err_msg=f"threshold_count T={T} cells={cells} {freq} {op} {thr}")
        red = str(rng.choice(["sum", "mean", "min", "max", "std", "var", "count"]))
        got = xgen.select_resample_op(x, red, ta, freq, device=dev)
        ref = ogen.select_resample_op(x, red, ot, freq)
        np.testing.assert_allclose(got, ref, rtol=1e-6, atol=1e-6, equal_nan=True, err_msg=f"resample {red} T={T} {cells} {freq}")


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_run_length_family(dev, seed):
    rng = np.random.default_rng(2000 + seed)
    for _ in range(6):
        T, cells, nanf, start = _case(rng)
        ta, ot = _axes(start, T)
        p_on = float(rng.choice([0.1, 0.5, 0.9]))
        m = (rng.random((T,) + cells) < p_on).astype(np.float32)
        m[rng.random(m.shape) < nanf] = np.nan
        freq = rng.choice(FREQS + [None])
        freq = None if freq is None else str(freq)
        window = int(rng.integers(1, 8))
        index = str(rng.choice(["first", "last"]))
        tag = f"T={T} cells={cells} nan={nanf} freq={freq} w={window} index={index}"
        for red in ("max", "min", "sum", "count", "mean", "std"):
            got = xrl.rle_statistics(m, red, window, freq=freq, time=ta, index=index, device=dev)
            ref = orl.rle_statistics(m, red, window, time=ot, freq=freq, index=index)
            np.testing.assert_allclose(got, ref, rtol=1e-6, atol=0, err_msg=f"rle_statistics {red} {tag}")
        np.testing.assert_array_equal(xrl.rle(m, index=index, device=dev), orl.rle(m, index=index), err_msg=f"rle {tag}")
        np.testing.assert_array_equal(xrl.windowed_run_count(m, window, freq=freq, time=ta, index=index, device=dev),
                                      orl.windowed_run_count(m, window, time=ot, freq=freq, index=index), err_msg=f"wrc {tag}")
        np.testing.assert_array_equal(xrl.windowed_run_events(m, window, freq=freq, time=ta, index=index, device=dev),
                                      orl.windowed_run_events(m, window, time=ot, freq=freq, index=index), err_msg=f"wre {tag}")
        mb = np.nan_to_num(m) > 0
        np.testing.assert_array_equal(xrl.first_run(mb, window, freq=freq, time=ta, device=dev), orl.first_run(mb, window, time=ot, freq=freq),
                                      err_msg=f"first_run {tag}")
        np.testing.assert_array_equal(xrl.last_run(mb, window, freq=freq, time=ta, device=dev), orl.last_run(mb, window, time=ot, freq=freq),
                                      err_msg=f"last_run {tag}")
        ws, wt = int(rng.integers(1, 6)), int(rng.integers(1, 6))
        np.testing.assert_array_equal(xrl.runs_with_holes(mb, ws, ~mb, wt, device=dev), orl.runs_with_holes(mb, ws, ~mb, wt), err_msg=f"holes {tag}")


@pytest.mark.parametrize("seed", range(10))
def test_fuzz_spells_and_percentiles(dev, seed):
    rng = np.random.default_rng(3000 + seed)
    for _ in range(4):
        T, cells, nanf, start = _case(rng)
        ta, ot = _axes(start, T)
        x = rng.gamma(0.6, 3.0, (T,) + cells).astype(np.float32) * (rng.random((T,) + cells) < 0.6)
        x = x.astype(np.float32)
        x[rng.random(x.shape) < nanf] = np.nan
        window = int(rng.integers(1, 7))
        red = str(rng.choice(["min", "max", "sum", "mean"]))
        op = str(rng.choice([">", ">=", "<", "<="]))
        thr = float(rng.choice([0.0, 1.0, 3.0]))
        gap = int(rng.choice([1, 1, 2, 4]))
        tag = f"T={T} cells={cells} w={window} {red} {op} {thr} gap={gap}"
        np.testing.assert_array_equal(xgen.spell_mask(x, window, red, op, thr, min_gap=gap, device=dev),
                                      ogen.spell_mask(x, window, red, op, thr, min_gap=gap), err_msg=f"spell_mask {tag}")
        freq = str(rng.choice(FREQS))
        for sr in ("max", "sum", "count"):
            before = bool(rng.integers(0, 2))
            np.testing.assert_array_equal(
                xgen.spell_length_statistics(x, thr, window, red, op, sr, ta, freq, min_gap=gap, resample_before_rl=before, device=dev),
                ogen.spell_length_statistics(x, thr, window, red, op, sr, ot, freq, before, gap), err_msg=f"sls {sr} {tag} {freq} {before}")
        if T >= 365:
            w = int(rng.choice([3, 5, 7, 9]))
            per = [float(v) for v in rng.choice([1.0, 10.0, 25.0, 50.0, 75.0, 90.0, 99.0], size=2, replace=False)]
            p = percentile_doy(x, ta, window=w, per=per, device=dev)
            exp, doys = ocal.percentile_doy(x, ot, w, per)
            np.testing.assert_allclose(p.values(), exp, rtol=1e-12, atol=0, equal_nan=True, err_msg=f"percentile_doy T={T} w={w} per={per}")


@pytest.mark.parametrize("seed", range(10))
def test_fuzz_seasons_events_quantiles(dev, seed):
    from oracle import sdba as osdba
    from xclim_amd import kernels as K

    rng = np.random.default_rng(4000 + seed)
    for _ in range(4):
        T, cells, nanf, start = _case(rng)
        ta, ot = _axes(start, T)
        p_on = float(rng.choice([0.2, 0.5, 0.8, 1.0, 0.0]))
        cond = rng.random((T,) + cells) < p_on
        window = int(rng.integers(1, 7))
        mid = rng.choice(["07-01", "01-15", "12-31", None])
        mid = None if mid is None else str(mid)
        tag = f"T={T} cells={cells} start={start} p_on={p_on} w={window} mid={mid}"
        # seasons per year (rl:891-1145); a period that does not contain the date gives NaN / 0
        got = xrl.season(cond, window, mid, time=ta, freq="YS", device=dev)
        es, ee, el = orl.season_per_period(cond, window, mid, ot, "YS")
        np.testing.assert_array_equal(got["start"], es, err_msg=f"season start {tag}")
        np.testing.assert_array_equal(got["end"], ee, err_msg=f"season end {tag}")
        np.testing.assert_array_equal(got["length"], el, err_msg=f"season length {tag}")
        # event tables (rl:1760-1901)
        ws, wt = int(rng.integers(1, 5)), int(rng.integers(1, 4))
        data = rng.gamma(2.0, 2.0, (T,) + cells).astype(np.float32)
        data[rng.random(data.shape) < nanf] = np.nan
        g = xrl.find_events(cond, ws, None, wt, data=data, device=dev)
        r = orl.find_events(cond, ws, None, wt, data=data)
        for k in r:
            np.testing.assert_allclose(g[k], r[k], rtol=1e-6, equal_nan=True, err_msg=f"find_events {k} {tag} ws={ws} wt={wt}")
        np.testing.assert_array_equal(xrl.run_bounds(cond, device=dev), orl.run_bounds(cond), err_msg=f"run_bounds {tag}")
        np.testing.assert_array_equal(xrl.keep_longest_run(cond, device=dev), orl.keep_longest_run(cond), err_msg=f"keep_longest {tag}")
        # full-series quantiles (E1) on the same shapes, both layouts
        x = rng.normal(0, 1, (T, int(np.prod(cells)))).astype(np.float32)
        x[rng.random(x.shape) < nanf] = np.nan
        nq = int(rng.choice([1, 5, 20, 50]))
        q = osdba.equally_spaced_nodes(nq)
        exp = osdba.quantile(x, q)
        np.testing.assert_allclose(K.quantile_series(dev, dev.to_device(x), q).get(), exp, rtol=1e-6, atol=0, equal_nan=True,
                                   err_msg=f"quantile_series {tag} nq={nq}")
        np.testing.assert_allclose(K.quantile_series(dev, dev.to_device(np.ascontiguousarray(x.T)), q, time_axis=1).get(), exp,
                                   rtol=1e-6, atol=0, equal_nan=True, err_msg=f"quantile_series T-minor {tag} nq={nq}")


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_multi_year_percentiles(dev, seed):
    rng = np.random.default_rng(5000 + seed)
    for _ in range(3):
        nyears = int(rng.choice([2, 3, 7, 12, 31]))
        cal = str(rng.choice(["standard", "noleap"]))
        start = str(rng.choice(["1980-01-01", "1991-07-01", "2000-03-01"]))
        T = int(365.25 * nyears) + int(rng.integers(-40, 40))
        cells = (int(rng.choice([1, 3, 8])), int(rng.choice([1, 5])))
        x = (280 + 10 * np.sin(np.arange(T)[:, None, None] / 58.0) + rng.normal(0, 3, (T,) + cells)).astype(np.float32)
        x[rng.random(x.shape) < float(rng.choice([0.0, 0.01, 0.2]))] = np.nan
        if cal == "standard":
            ta, ot = TimeAxis.daily(start, T, "standard"), OTime.standard(start, T)
        else:
            if not start.endswith("01-01"):
                start = start[:4] + "-01-01"  # (the oracle's noleap axis starts on Jan 1)
            ta, ot = TimeAxis.daily(start, T, "noleap"), OTime.noleap(int(start[:4]), T)
        w = int(rng.choice([3, 5, 7, 9]))
        per = sorted(float(v) for v in rng.choice([1.0, 5.0, 10.0, 50.0, 90.0, 95.0, 99.0], size=int(rng.integers(1, 4)), replace=False))
        p = percentile_doy(x, ta, window=w, per=per, device=dev)
        exp, doys = ocal.percentile_doy(x, ot, w, per)
        np.testing.assert_array_equal(p.dayofyear, doys)
        np.testing.assert_allclose(p.values(), exp, rtol=1e-12, atol=0, equal_nan=True,
                                   err_msg=f"percentile_doy nyears={nyears} cal={cal} start={start} T={T} w={w} per={per}")


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_eqm_and_two_variable_reductions(dev, seed):
    from oracle import sdba as osdba
    from xclim_amd import kernels as K
    from xclim_amd.calendar import climatological_mean_doy

    rng = np.random.default_rng(6000 + seed)
    for _ in range(3):
        T, cells, nanf, start = _case(rng)
        if T < 20:
            T = 120
        ta, ot = _axes(start, T)
        C = int(np.prod(cells))
        ref = rng.normal(10, 3, (T, C)).astype(np.float32)
        hist = (rng.normal(11, 4, (T, C))).astype(np.float32)
        sim = (rng.normal(12, 4, (T, C))).astype(np.float32)
        for a in (ref, hist, sim):
            a[rng.random(a.shape) < nanf * 0.3] = np.nan
        nq = int(rng.choice([5, 10, 20, 32]))
        kind = str(rng.choice(["+", "*"]))
        eaf, ehq = osdba.eqm_train(ref, hist, nq, kind)
        af, hq = K.eqm_train(dev, dev.to_device(ref), dev.to_device(hist), osdba.equally_spaced_nodes(nq), kind)
        np.testing.assert_allclose(hq.get(), ehq, rtol=1e-6, equal_nan=True, err_msg=f"hist_q T={T} C={C} nq={nq}")
        eaf32, ehq32 = eaf.astype(np.float32), ehq.astype(np.float32)
        for interp in ("nearest", "linear", "cubic"):
            if interp == "cubic" and (nq < 4 or np.isnan(ehq32).any() or (np.diff(ehq32, axis=0) <= 0).any()):
                continue  # scipy's cubic needs >= 4 strictly increasing nodes
            extrap = str(rng.choice(["constant", "nan"]))
            got = K.eqm_adjust(dev, dev.to_device(sim), dev.to_device(eaf32), dev.to_device(ehq32), kind, interp, extrap).get()
            exp = osdba.eqm_adjust(sim, eaf32, ehq32, kind, interp, extrap)
            # north star: 1e-6 relative for the piecewise-constant / piecewise-linear factors; the cubic spline
            # (scipy fp64 in the oracle, fp32 Horner on the device) is held to 2e-6
            np.testing.assert_allclose(got, exp, rtol=2e-6 if interp == "cubic" else 1e-6, atol=0, equal_nan=True,
                                       err_msg=f"adjust {interp} {extrap} {kind} T={T} C={C} nq={nq}")
        # quantile delta mapping on the same factors: exact average ranks + fp64 lookup (oracle: scipy rankdata / interp1d)
        for interp in ("nearest", "linear"):
            extrap = str(rng.choice(["constant", "nan"]))
            got = K.qdm_adjust(dev, dev.to_device(sim), dev.to_device(eaf32), osdba.equally_spaced_nodes(nq), kind, interp, extrap).get()
            exp = osdba.qdm_adjust(sim, eaf32, osdba.equally_spaced_nodes(nq), kind, interp, extrap)
            np.testing.assert_allclose(got, exp, rtol=1e-6, atol=0, equal_nan=True, err_msg=f"qdm {interp} {extrap} {kind} T={T} C={C} nq={nq}")
        # two-variable range reductions
        lo = ref.reshape((T,) + cells)
        hi = (ref + np.abs(hist)).reshape((T,) + cells)
        freq = str(rng.choice(FREQS))
        for red in ("max", "min", "mean", "sum"):
            np.testing.assert_allclose(xgen.diurnal_temperature_range(lo, hi, red, ta, freq, device=dev),
                                       ogen.diurnal_temperature_range(lo, hi, red, ot, freq), rtol=1e-6, atol=1e-5 if red == "sum" else 0,
                                       equal_nan=True, err_msg=f"dtr {red} T={T} {cells} {freq}")
        np.testing.assert_allclose(xgen.interday_diurnal_temperature_range(lo, hi, ta, freq, device=dev),
                                   ogen.interday_diurnal_temperature_range(lo, hi, ot, freq), rtol=1e-6, equal_nan=True)
        np.testing.assert_allclose(xgen.extreme_temperature_range(lo, hi, ta, freq, device=dev),
                                   ogen.extreme_temperature_range(lo, hi, ot, freq), rtol=1e-6, equal_nan=True)
        if T >= 365:
            w = int(rng.choice([3, 5, 7, 9]))
            m, s, doys = climatological_mean_doy(lo, ta, window=w, device=dev)
            em, es, ed = ocal.climatological_mean_doy(lo, ot, w)
            np.testing.assert_array_equal(doys, ed)
            np.testing.assert_allclose(m, em, rtol=1e-6, equal_nan=True, err_msg=f"doy mean T={T} w={w}")
            np.testing.assert_allclose(s, es, rtol=2e-6, atol=1e-6, equal_nan=True, err_msg=f"doy std T={T} w={w}")
