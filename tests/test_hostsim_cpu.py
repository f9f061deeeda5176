"""Kernel parity WITHOUT a GPU (tests/hostsim — test infrastructure only).  The translation units of xclim_amd/csrc are compiled
with g++ against a stand-in for the HIP runtime and run on the CPU:
  * thread by thread — detrend, window, runlen, reduce, reduce2, spell, elemwise, eqm, wquantile (no LDS traffic between threads),
    and plane.hip with its wave-aggregated work-list appends as waves of one lane;
  * every workgroup as a set of FIBERS (simt.h: __syncthreads, wave-uniform shuffles / votes / readlane, atomics) — f64, select,
    select5, tcount, qdm, quantile, doystats and the kernels of core.hip (transposes, synthetic fields).
63 of the 93 entry points of include/xclim_hip.h exist in that build: every compute entry point but xh_adapt_freq (rocPRIM); the
other 30 are runtime services (memory, streams, RCCL).  The kernels written at ISA level (register sorting networks, DPP, the
streaming selection of select4.hip, the register top-16 percentile kernels) are NOT simulated: their launchers answer "not this
kernel's shape" and the callers' general kernels run — exactly the fall-back the product takes for shapes those kernels decline.
The SAME parity tests the GPU runs (the functions of tests/test_gpu_kernels.py, test_gpu_spells.py, test_gpu_plane.py, ...) are
called with the simulated device on a subset of their parameters.  The `-m gpu` runs remain the parity tests proper (the real
kernels on the real device, the ISA-level ones included); the product has no CPU path: the simulation library is built into a
temporary directory by this module only, and what is not simulated raises instead of pretending."""
import numpy as np
import pytest

from oracle import sdba as osdba
from xclim_amd import kernels as K


@pytest.fixture(scope="module")
def sim(tmp_path_factory):
    from tests.hostsim import simdevice

    import subprocess

    try:
        path = simdevice.build(str(tmp_path_factory.mktemp("hostsim")))
    except (RuntimeError, subprocess.CalledProcessError) as e:   # (no g++ / a g++ that does not take the stand-in: not a product failure)
        pytest.skip(f"host simulation not built here: {e}")
    return simdevice.SimDevice(path)


@pytest.fixture(scope="module")
def tk():
    import tests.test_gpu_kernels as mod

    return mod


def test_unsimulated_entry_points_raise(sim):
    with pytest.raises(NotImplementedError, match="not simulated"):
        K.doy_mean_std(sim, sim.to_device(np.zeros((730, 4), np.float32)), np.arange(730, dtype=np.int32).reshape(2, 365), 5)


@pytest.mark.parametrize("op", [">", "<", ">=", "<=", "==", "!="])
def test_threshold_count(sim, rng, tk, op):
    tk.test_threshold_count_scalar(sim, rng, 7, op)


def test_threshold_count_other_forms(sim, rng, tk):
    tk.test_threshold_count_scalar_promotion(sim, rng)
    tk.test_threshold_count_doy_and_full(sim, rng, 3)      # (the multi-year tile kernel is not simulated: its documented fall-back runs)
    tk.test_domain_count(sim, rng)


@pytest.mark.parametrize("reducer", ["sum", "mean", "min", "max", "std", "var", "count", "argmin", "argmax"])
def test_resample_reduce(sim, rng, tk, reducer):
    tk.test_resample_reduce(sim, rng, reducer, 5)


@pytest.mark.parametrize("window,center,reducer", [(5, True, "sum"), (3, False, "mean"), (14, True, "max"), (4, True, "min"), (5, True, "std")])
def test_rolling_reduce(sim, rng, tk, window, center, reducer):
    tk.test_rolling_reduce(sim, rng, window, center, reducer)


@pytest.mark.parametrize("index", ["first", "last"])
def test_run_length_family(sim, rng, tk, index):
    tk.test_cumsum_reset_and_rle(sim, rng, index, 0.05)
    for stat in ("max", "min", "sum", "count", "mean", "std"):
        tk.test_run_stats_mask(sim, rng, stat, index, True)
    tk.test_run_stats_mask(sim, rng, "max", index, False)
    tk.test_windowed_run_count_events(sim, rng, 2, index)


def test_boundary_runs_and_fused_threshold(sim, rng, tk):
    tk.test_first_last_run(sim, rng, 3, True)
    tk.test_first_last_run(sim, rng, 7, False)
    tk.test_cdd_fused(sim, rng)


def test_detrend_pieces(sim):
    rng = np.random.default_rng(9)
    T = 500
    x = (283 + rng.normal(0, 4, (T, 7)) + 0.01 * np.arange(T)[:, None]).astype(np.float32)
    x[rng.random(x.shape) < 0.05] = np.nan
    x[:, 3] = np.nan
    d = sim.to_device(x)
    for w in (1, 5, 31):
        np.testing.assert_allclose(K.window_nanmean(sim, d, w).get(), osdba.window_nanmean(x, w), rtol=3e-7, atol=1e-6, equal_nan=True)
    p0, p1 = K.poly_trend(sim, d, 1)
    trend = osdba.poly_trend(x, 1)
    t = np.arange(T)[:, None] - 0.5 * (T - 1)
    np.testing.assert_allclose(p0.get()[None, :] + p1.get()[None, :] * t, trend, rtol=1e-9, atol=1e-9, equal_nan=True)
    detr = K.trend_apply(sim, d, p0, p1, "-").get()
    np.testing.assert_allclose(detr, (x.astype(np.float64) - trend).astype(np.float32), rtol=1e-6, atol=1e-5, equal_nan=True)


@pytest.fixture(scope="module")
def ts():
    import tests.test_gpu_spells as mod

    return mod


@pytest.mark.parametrize("window,red,op", [(3, "min", ">"), (3, "max", "<="), (4, "sum", ">="), (5, "mean", ">")])
def test_spell_mask(sim, rng, ts, window, red, op):
    ts.test_spell_mask(sim, rng, window, red, op, 0.02)


def test_spells_and_seasons(sim, rng, ts):
    ts.test_spell_mask_weights_and_gap(sim, rng)
    ts.test_spell_length_statistics_general(sim, rng, 3, "min", "max", True)
    ts.test_spell_length_statistics_general(sim, rng, 2, "max", "sum", False)
    ts.test_reference_spell_length_statistics_answer(sim)
    ts.test_runs_with_holes(sim, rng, 2, 3)
    ts.test_keep_longest_run(sim, rng)
    ts.test_season(sim, rng, 3, "07-01")
    ts.test_season(sim, rng, 1, None)
    ts.test_reference_season_answers(sim)
    ts.test_date_bounded_runs(sim, rng, 2, "07-01")
    ts.test_windowed_max_run_sum(sim, rng, 3)
    ts.test_run_bounds(sim, rng)
    ts.test_find_events(sim, rng, 2, 1, "MS")
    ts.test_suspicious_run(sim, rng, 3, ">", None)
    ts.test_suspicious_run(sim, rng, 1, "==", 2.0)


def test_spells_of_two_variables_and_doy_thresholds(sim, rng, ts):
    ts.test_spell_mask_two_variables(sim, rng, 3, "min", ">=", "all")
    ts.test_spell_mask_two_variables(sim, rng, 2, "sum", "<=", "any")
    ts.test_bivariate_spell_length_statistics_and_thresholded_events(sim, rng)
    ts.test_run_stats_doy_fused(sim, rng, 730, 37, ">")
    ts.test_run_stats_doy_fused(sim, rng, 1461, 5, "!=")
    ts.test_spell_length(sim, rng, "max", ">")
    ts.test_1d_variants_and_season_end(sim, rng)


@pytest.mark.parametrize("kind", ["+", "*"])
@pytest.mark.parametrize("interp", ["nearest", "linear", "cubic"])
def test_eqm_adjust(sim, rng, tk, kind, interp):
    """E2 of the hot path (xh_eqm_adjust): the node search / interpolation kernels against scipy's interp1d, from the ORACLE's
    node tables, incl. NaN nodes and NaN samples."""
    T, C = 400, 40
    ref, hist = tk._field(rng, T, C), (tk._field(rng, T, C) + 1.5).astype(np.float32)
    sim_x = (tk._field(rng, T, C, nan_frac=0.01) + 2.0).astype(np.float32)
    hist[:, 0] = np.nan
    eaf, ehq = osdba.eqm_train(ref, hist, 20, kind)
    eaf, ehq = eaf.astype(np.float32), ehq.astype(np.float32)
    eaf[1, 3] = np.nan
    ehq[18, 5] = np.nan
    for extrap in ("constant", "nan"):
        got = K.eqm_adjust(sim, sim.to_device(sim_x), sim.to_device(eaf), sim.to_device(ehq), kind, interp, extrap).get()
        exp = osdba.eqm_adjust(sim_x, eaf, ehq, kind, interp, extrap)
        np.testing.assert_allclose(got, exp, rtol=2e-6 if interp == "cubic" else 1e-6, atol=0, equal_nan=True, err_msg=extrap)


def test_apply_factor(sim, rng, tk):
    """utils.apply_correction on two fields (QDM "cubic": the factor comes out of an interpolation over the ranks)"""
    x = tk._field(rng, 50, 9)
    f = rng.normal(1.0, 0.1, (50, 9)).astype(np.float32)
    np.testing.assert_array_equal(K.apply_factor(sim, sim.to_device(x), sim.to_device(f), "*").get(), x * f)
    np.testing.assert_array_equal(K.apply_factor(sim, sim.to_device(x), sim.to_device(f), "+").get(), x + f)


@pytest.fixture(scope="module")
def tp():
    import tests.test_gpu_plane as mod

    return mod


@pytest.mark.parametrize("G,nq", [(12, 20), (12, 5), (40, 8), (365, 6)])
@pytest.mark.parametrize("fractional", [True, False])
def test_plane_linear_all_routes(sim, rng, tp, G, nq, fractional):
    """xh_plane_linear through ALL of its kernels — the row kernel (integer coordinates), the pair kernel (fractional ones, <= 20
    nodes), their work lists and the Delaunay walk — against the real scipy.griddata, for narrow and wide node spacings."""
    for scale, kind in ((0.05, "t"), (6.0, "t"), (12.0, "p")):
        tp.test_plane_linear_matches_griddata(sim, rng, G, nq, scale, kind, fractional)


def test_plane_linear_special_nodes_and_nearest(sim, rng, tp):
    tp.test_plane_linear_nan_nodes_and_ties(sim, rng)
    tp.test_plane_linear_kinds_and_base(sim, rng)
    # xh_plane_nearest (the row kernel's NEAREST form + the listed neighbours) against griddata "nearest"
    G, nq, C, T = 12, 9, 4, 300
    xq = np.sort(rng.gamma(0.7, 12.0, (G, nq, C)), axis=1).astype(np.float32)
    yq = rng.normal(0, 1, (G, nq, C)).astype(np.float32)
    x = rng.uniform(0, float(xq.max()) * 1.05, (T, C)).astype(np.float32)
    x[rng.random((T, C)) < 0.03] = np.nan
    g = rng.integers(1, G + 1, T).astype(np.float64)
    for extrap in ("constant", "nan"):
        got = K.plane_nearest(sim, sim.to_device(x), g, sim.to_device(yq), sim.to_device(xq), "factor", extrap).get()
        exp = osdba.interp_on_quantiles_2d(x, g, np.arange(1, G + 1), xq, yq, "nearest", extrap)
        assert np.array_equal(np.isnan(got), np.isnan(exp))
        assert (~np.isclose(got, exp, rtol=1e-6, equal_nan=True)).sum() <= 2   # (an exact tie between two rows may go either way)


@pytest.mark.parametrize("R", [3, 40])
def test_weighted_ensemble_percentiles(sim, rng, R):
    """xh_weighted_quantile (wquantile.hip: insertion sort in lane-private LDS columns) against the restated xarray estimator."""
    import tests.test_gpu_api as ta

    ta.test_weighted_ensemble_percentiles(sim, rng, R)


# ---- kernels that talk through LDS and the wave, on fibers (tests/hostsim/simt.h) --------------------------------------------------
@pytest.fixture(scope="module")
def tf():
    import tests.test_gpu_f64 as mod

    return mod


@pytest.mark.parametrize("N", [1, 7, 150, 930])
def test_float64_quantiles_on_fibers(sim, rng, tf, N):
    """xh_nan_quantile_f64 (f64.hip: a workgroup per column, LDS, __syncthreads, __shfl_xor) — every thread a fiber."""
    tf.test_calc_perc_float64_vs_oracle(sim, rng, N)


def test_float64_counts_and_reductions(sim, rng, tf):
    tf.test_calc_perc_float64_matches_the_reference_bitwise(sim)
    tf.test_threshold_count_float64_counts_exactly(sim, rng)
    tf.test_select_resample_op_float64(sim, rng)


@pytest.fixture(scope="module")
def tapi():
    import tests.test_gpu_api as mod

    return mod


@pytest.mark.parametrize("T,C", [(40, 7), (365, 100), (1000, 33), (5000, 6)])
def test_quantile_series_on_fibers(sim, rng, tk, T, C):
    """xh_quantile_series (E1): 8-lane groups on time-major rows (T <= 512), transposes + one wave per column with the column in
    LDS (<= 1024), 1024-thread workgroups beyond (the register-sort, two-pass-histogram and lean kernels are ISA-level: their
    callers' general kernels run instead)."""
    tk.test_quantile_series(sim, rng, T, C)


def test_quantile_series_hard_distributions_on_fibers(sim, rng, tk):
    for T in (40, 365, 600):
        for kind in ("pr", "pr_skewed", "two_values", "negative_floor", "clustered", "constant", "nan_heavy"):
            tk.test_quantile_series_hard_distributions(sim, rng, T, kind)
    tk.test_quantile_series_beyond_32768_steps(sim, rng, 40000)        # the radix select of select5.hip


@pytest.mark.parametrize("kind,interp,extrap", [("+", "nearest", "constant"), ("*", "linear", "nan")])
def test_eqm_train_and_adjust_on_fibers(sim, rng, tk, kind, interp, extrap):
    tk.test_eqm_train_adjust(sim, rng, kind, interp, extrap)
    tk.test_eqm_precipitation_tied_nodes(sim, rng, kind, interp, extrap)


def test_percentiles_on_fibers(sim, rng, tk):
    """calc_perc (xh_nan_quantile: register bitonic, lane-private LDS columns), percentile_doy on one year (sliding window) and on
    three (sorted day-set lists in an LDS ring), through a virtual time map, the 366-day re-gridding, +-inf samples."""
    for N, C in ((1, 10), (2, 10), (5, 300), (13, 70), (30, 129), (150, 200), (365, 40)):
        tk.test_nan_quantile(sim, rng, N, C, (1.0, 1.0))
    tk.test_nan_quantile(sim, rng, 600, 70, (1 / 3, 1 / 3))
    tk.test_percentile_doy(sim, rng, 1, 5, 260, "noleap")
    tk.test_percentile_doy(sim, rng, 3, 5, 33, "standard")
    tk.test_percentile_doy_virtual_time_map(sim, rng, 4)
    tk.test_doy_interp(sim, rng)
    tk.test_infinities_follow_the_nanmax_rule(sim, rng)


def test_threshold_count_doy_tile_kernel_on_fibers(sim, rng, tk):
    """xh_threshold_count_doy on multi-year series: the LDS tile kernel of tcount.hip (the 30-year tx90p count of the headline)."""
    tk.test_threshold_count_doy_multi_year_tile_kernel(sim, rng, ">", 200, "YS")
    tk.test_threshold_count_doy_multi_year_tile_kernel(sim, rng, "<=", 64, "MS")
    tk.test_threshold_count_doy_tile_kernel_empty_periods_and_narrow_counters(sim, rng)


@pytest.mark.parametrize("kind", ["+", "*"])
@pytest.mark.parametrize("interp", ["nearest", "linear", "cubic"])
def test_qdm_adjust_on_fibers(sim, rng, tapi, kind, interp):
    """QuantileDeltaMapping.adjust through the exact-rank kernel (qdm.hip: a column per workgroup, average ranks through LDS); the
    one-year cut-value kernel and the streaming kernels are ISA-level and decline in the simulation."""
    for T, cells in ((365, (7, 9)), (800, (33,)), (50, (4, 4)), (1, (3,))):
        tapi.test_qdm_adjust_matches_oracle(sim, rng, kind, interp, T, cells)


def test_synthetic_fields_and_transposes_on_fibers(sim, rng, tk):
    tk.test_synthetic_matches_oracle(sim)
    for shape in ((130, 77), (128, 128), (300, 388), (257, 260), (5, 4), (8, 4)):
        tk.test_transpose(sim, rng, shape)
