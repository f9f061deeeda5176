"""Kernel parity WITHOUT a GPU (tests/hostsim — test infrastructure only).  The translation units of xclim_amd/csrc are compiled
with g++ against a stand-in for the HIP runtime and run on the CPU:
  * thread by thread — detrend, window, runlen, reduce, spell, elemwise, eqm, wquantile (no LDS traffic between threads),
    and plane.hip with its wave-aggregated work-list appends as waves of one lane;
  * every workgroup as a set of FIBERS (simt.h: __syncthreads, wave-uniform shuffles / votes / readlane, atomics) — f64, select,
    select2, select3, select4, select5, tcount, qdm, qdm2, quantile, doystats, reduce2, pdoy_top, pdoy_quad, pdoy_walk, winsel and the
    kernels of core.hip (transposes, synthetic fields).
70 of the 100 entry points of include/xclim_hip.h exist in that build: every compute entry point but xh_adapt_freq (rocPRIM); the
other 30 are runtime services (memory, streams, RCCL).  The register percentile kernels (pdoy_top / pdoy_quad / pdoy_walk) run on
fibers too, with the four ISA statements of topnet.h rewritten to the C++ they stand for, and so do the register sorting networks
(select3 / qdm2: the DPP split across the lane pair as a shuffle), select2's wave counts on VCC and the streaming two-pass selection
of select4.hip (DPP lane exchanges as shuffles, the execution-masked LDS appends as conditional stores, v_readlane from divergent
code as publish + peek).  rocPRIM (qdm3.hip: the global sort behind xh_adapt_freq and the exact-rank fall-back of series beyond
32768 steps) and RCCL are NOT simulated: those launchers answer "not built" and the tests that need them stay GPU-only.
The SAME parity tests the GPU runs are re-run here: whole modules of the `-m gpu` suite (edges, patch, api, spells, f64, kernels,
plane: ~740 tests, minus the slow sizes) in two child pytest runs against the simulation library, selected tests of the
ISA-level kernels in three more, plus a few direct calls.  The `-m gpu` runs remain the parity tests proper (the real
kernels on the real device, the ISA-level ones included); the product has no CPU path: the simulation library is built into a
temporary directory by this module only, and what is not simulated raises instead of pretending."""
import os

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
    """(xh_adapt_freq ranks through rocPRIM's segmented sort; the RCCL entry points are runtime services)"""
    for name in ("xh_adapt_freq", "xh_comm_allgather"):
        with pytest.raises(NotImplementedError, match="not simulated"):
            getattr(sim.lib, name)


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


@pytest.fixture(scope="module")
def tapi():
    import tests.test_gpu_api as mod

    return mod


@pytest.mark.parametrize("kind", ["+", "*"])
@pytest.mark.parametrize("interp", ["nearest", "linear", "cubic"])
def test_qdm_adjust_on_fibers(sim, rng, tapi, kind, interp):
    """QuantileDeltaMapping.adjust: the one-year cut-value kernel (qdm2.hip) for "nearest", the exact-rank kernel (qdm.hip: a column
    per workgroup, average ranks through LDS) for the interpolating modes."""
    for T, cells in ((365, (7, 9)), (800, (33,)), (50, (4, 4)), (1, (3,))):
        tapi.test_qdm_adjust_matches_oracle(sim, rng, kind, interp, T, cells)


def _child_run(sim, files, skip="nothing_is_skipped", deselect=(), at_least=1):
    """The given modules of the `-m gpu` suite in a child pytest against the simulation library built for this module."""
    import os
    import subprocess
    import sys

    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    env = dict(os.environ, XH_TEST_DEVICE="hostsim", HOSTSIM_LIB=sim.path)
    cmd = [sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-p", "no:cacheprovider", "--timeout", "600", *files, "-k", f"not ({skip})"]
    try:   # (the tests are independent and seeded one by one: four workers where pytest-xdist is installed)
        import xdist  # noqa: F401

        cmd += ["-n", str(min(4, os.cpu_count() or 1))]
    except ImportError:
        pass
    for d in deselect:
        cmd += ["--deselect", d]
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True)
    tail = res.stdout.strip().splitlines()[-1] if res.stdout.strip() else res.stderr[-500:]
    assert res.returncode == 0, res.stdout[-3000:]
    assert " passed" in tail and "failed" not in tail, tail
    assert int(tail.split(" passed")[0].split()[-1]) >= at_least, tail


def test_whole_gpu_modules_on_the_simulation(sim):
    """The host mirrors of the index functions — the Python the GPU path runs: indices.py, calendar.py, run_length.py, generic.py,
    missing.py, the xarray-facing adapter and patch — through the simulated kernels: whole modules of the `-m gpu` suite (edge
    cases: ragged / empty grids, single steps, all-NaN, error codes; the apply_ufunc views; the index-level API incl. the reference's
    known answers, tx90p / tx10p at Indicator level, percentile_doy with the 366-day re-gridding; every spell / season / run-length
    test; the float64 kernels; the xarray-facing adapter on the duck-typed DataArray; the block adapter's slab pipeline).  What needs the ISA-level kernels is deselected."""
    skip = ("qdm or eqm or dqm or sdba or bootstrap or adapt or add_dims or sub_groupings or exceedance_fused or "
            "beyond or grouped or plane or quantile_cells or tx90p_on_a_float64 or refused_elsewhere")
    _child_run(sim, ["tests/test_gpu_edges.py", "tests/test_gpu_patch.py", "tests/test_gpu_api.py", "tests/test_gpu_spells.py",
                     "tests/test_gpu_f64.py", "tests/test_gpu_adapter.py", "tests/test_gpu_blocks.py"], skip, at_least=270,
               # (the device-resident input cache is the real Device's; the simulation's buffers are host memory)
               deselect=["tests/test_gpu_adapter.py::test_inputs_and_tables_stay_on_the_device_across_wrapper_calls"])


def test_selection_percentile_and_plane_modules_on_the_simulation(sim):
    """tests/test_gpu_kernels.py and tests/test_gpu_plane.py: every entry point against the oracle — counts, reductions, rolling,
    run lengths, calc_perc, percentile_doy (one year; multi-year through the LDS ring for the middle of the distribution), quantile
    series of 40 ... 55 152 steps incl. the hard distributions, EQM train / adjust (nearest, linear, cubic), the plane kernels.
    Deselected here (and run, a few parameter sets each, by the three tests below): the tests of the register sorts, the two-pass
    histogram, the register top-16 / quad / walk percentile kernels; and the many-column sizes."""
    skip = ("many_columns or percentile_doy_quad or percentile_doy_count_multi_year or percentile_doy_merge_path or walk_kernel or "
            "virtual_time_map or register_sort or two_pass")
    if not os.environ.get("HOSTSIM_SLOW"):   # (40 000 / 55 152 steps through select4's collect rounds + the radix select: 60 s here)
        skip += " or beyond_32768"
    _child_run(sim, ["tests/test_gpu_kernels.py", "tests/test_gpu_plane.py"], skip, at_least=290,
               deselect=["tests/test_gpu_kernels.py::test_percentile_doy[9-7-20-standard]",
                         "tests/test_gpu_kernels.py::test_percentile_doy[30-5-24-noleap]"])


def test_register_percentile_kernels_on_the_simulation(sim):
    """The multi-year percentile kernels of the headline chain — the register top-16 kernel (pdoy_top.hip), the quad kernel of the
    30-year tx90p (pdoy_quad.hip), the split-walk kernel for the middle of the distribution (pdoy_walk.hip), the fused count, the
    LDS-ring merge — on fibers: readlane row fetches, buffer loads and the comparator networks of topnet.h (whose four ISA
    statements — v_min_f32, v_max_f32, the NaN-replace-and-count triples — are rewritten to the C++ they stand for).  A few
    parameter sets of each GPU test; bootstrap and the fused exceedance count ride on the same kernels."""
    k = "tests/test_gpu_kernels.py::"
    a = "tests/test_gpu_api.py::"
    ids = [k + "test_percentile_doy[9-7-20-standard]", k + "test_percentile_doy[30-5-24-noleap]",
           k + "test_percentile_doy_merge_path[12-7-standard-0.05]", k + "test_percentile_doy_quad_kernel[30-noleap-90.0-0.0]",
           k + "test_percentile_doy_count_multi_year[YS-90.0->]", k + "test_percentile_doy_virtual_time_map[12]",
           a + "test_percentile_bootstrap[noleap-2000-01-01-7-base0-YS]", a + "test_percentile_exceedance_fused[noleap-4380-YS-5-95.0->]"]
    if os.environ.get("HOSTSIM_SLOW"):   # (the walk kernel exchanges through the wave at every step of every day: 50 s here)
        ids += [k + "test_percentile_doy_walk_kernel[9-5-noleap-0.9]", k + "test_percentile_doy_quad_kernel[7-noleap-5.0-0.0]"]
    _child_run(sim, ids, at_least=len(ids))


def test_register_sorting_network_kernels_on_the_simulation(sim):
    """The one-year kernels of the headline's EQM / QDM legs — select3.hip (k_select_regsort) and qdm2.hip (k_qdm_regsort +
    k_cut_classify): a column in the registers of a lane pair, the comparator networks of sortnet_183.h, the split across the pair
    by DPP.  On fibers, with the two DPP macros (xor with the lane's sign mask, the partner's complement, keep the larger) and qdm2's
    NaN-count triple rewritten to the C++ they stand for.  The GPU tests of these kernels, incl. bit-equality with the histogram /
    exact-rank kernels under the diagnostic switches."""
    k, a = "tests/test_gpu_kernels.py::", "tests/test_gpu_api.py::"
    _child_run(sim, [k + "test_quantile_series_one_year_register_sort", k + "test_quantile_series_register_sort_matches_histogram_kernels",
                     a + "test_qdm_nearest_one_year_cut_value_kernel", a + "test_qdm_precipitation_and_edge_cases"], at_least=30)


def test_streaming_selection_kernels_on_the_simulation(sim):
    """select4.hip — the two streaming passes behind the 30-year EQM training and the long-series QDM "nearest" (k_hs_sample,
    k_hs_hist, k_hs_collect in both modes, the wave-wide bitonic sort of the candidates, k_cut_classify) — and select2.hip's lean
    kernel, on fibers.  Rewritten for the host (simdevice.py, UNIT_REWRITES): the DPP lane exchanges of the bitonic network
    (a shuffle), v_med3_i32 on the bits of a float difference (a clamp), v_min / v_max / v_min3 / v_max3 (fminf / fmaxf), the
    sixteen LDS writes under the execution mask (conditional stores), the records hs_qdm_pick reads with v_readlane from
    per-lane searches (published once, peeked after).  Small T of each GPU test; test_quantile_series_register_sort_matches_
    histogram_kernels (other child run) now compares the register sort with THESE kernels bit for bit."""
    k, a = "tests/test_gpu_kernels.py::", "tests/test_gpu_api.py::"
    ids = [k + f"test_quantile_series_two_pass_histogram[{nq}-{T}]" for nq in (1, 20, 32) for T in (1025, 1300, 4000)]
    ids += [k + "test_quantile_series_two_pass_value_classes"]
    ids += [a + f"test_qdm_adjust_matches_oracle[{T}-cells{i}-nearest-{kind}]" for T, i in ((3000, 2), (10950, 3)) for kind in "+*"]
    _child_run(sim, ids, at_least=len(ids))


def test_quantile_mapping_api_on_the_simulation(sim):
    """The sdba half of tests/test_gpu_api.py — EmpiricalQuantileMapping / QuantileDeltaMapping / DetrendedQuantileMapping through
    the host mirror (xclim_amd/sdba.py: Grouper, month / season groupings, detrending, the 2-D interpolation over the
    (quantile, group) plane) and every kernel behind it, the ISA-level ones included.  Left to the GPU: adapt_freq (rocPRIM),
    series beyond 32768 steps (rocPRIM for the flagged columns), and the slow ones here (day-of-year groupings, bootstrap,
    20 000+ steps)."""
    _child_run(sim, ["tests/test_gpu_api.py"], at_least=60,
               skip="not (qdm or eqm or dqm or sdba) or bootstrap or adapt or sub_groupings or beyond or dayofyear or 32768 or 20000 "
                    "or sliding or small_groups or doy_training_without_window")   # (winsel.hip, k_qdm_groups: their own tests below, small shapes)


def test_sliding_window_training_on_the_simulation(sim, rng):
    """winsel.hip (round 6: the sorted sliding window of the day-of-year training) on fibers — DPP lane exchanges as shuffles —
    against xh_eqm_train on every group's gathered sample, bit for bit: the first 40 steps of a 3-year series with a window of 7
    days (NaN samples, ties, an infinity, an empty cell, a window that reaches beyond the start of the series).  The whole
    schedule, larger windows and 30 years: tests/test_gpu_api.py::test_eqm_doy_window_sliding_matches_per_group on the GPU."""
    from xclim_amd import sdba as xsdba
    from xclim_amd.timeaxis import TimeAxis

    T, cells, nq = 365 * 3, 5, 6
    ta = TimeAxis.daily("2001-01-01", T, "noleap")
    t = np.arange(T)[:, None]
    ref = np.round(288 + 10 * np.sin(2 * np.pi * t / 365) + rng.normal(0, 3, (T, cells)), 1).astype(np.float32)
    hist = (ref[::-1] * 1.01 + rng.normal(0, 1, (T, cells))).astype(np.float32)
    ref[rng.random(ref.shape) < 0.04] = np.nan
    hist[:, 3] = np.nan
    ref[2:5, 1] = np.inf
    rows0, enter, leave = xsdba.Grouper("time.dayofyear", 7).ring_schedule(ta)
    G = 41
    q = (np.arange(nq) + 0.5) / nq
    d_ref, d_hist = sim.to_device(ref), sim.to_device(hist)
    af, hq = K.eqm_train_window(sim, d_ref, d_hist, rows0, enter[:G - 1], leave[:G - 1], q, "+")
    af, hq = af.get(), hq.get()
    rows = rows0.copy()
    for g in range(G):
        a_g, h_g = K.eqm_train(sim, K.select_rows(sim, d_ref, rows), K.select_rows(sim, d_hist, rows), q, "+")
        np.testing.assert_array_equal(hq[g], h_g.get(), err_msg=f"group {g}")
        np.testing.assert_array_equal(af[g], a_g.get(), err_msg=f"group {g}")
        if g + 1 < G:   # one row per year leaves, one enters (the order of a sample's rows does not matter)
            for y in range(enter.shape[1]):
                hit = np.nonzero(rows == leave[g, y])[0]
                rows[hit[0] if leave[g, y] >= 0 and len(hit) else np.nonzero(rows < 0)[0][0]] = enter[g, y]
    assert np.isnan(hq[:, :, 3]).all() and np.isfinite(hq[:, :, 0]).all()


@pytest.mark.parametrize("kind", ["+", "*"])
def test_sliding_window_dqm_training_on_the_simulation(sim, rng, kind):
    """xh_dqm_train_window (winsel.hip with the normalisation of dqm_train) against the per-group chain it replaces — xh_poly_trend
    (degree 0), xh_trend_apply, xh_eqm_train on every group's gathered sample: the means to summation order, the tables bit for
    bit wherever the means agree in all their bits (checked: nearly everywhere) and to one part in 10^6 elsewhere.  Cells: plain,
    NaN samples, an infinity in the window (every finite sample becomes -inf / 0, the infinite one NaN: the window is
    normalised and sorted again), a negative mean ("*": the order reverses), all zeros ("*": 0 / 0), no valid sample."""
    from xclim_amd import sdba as xsdba
    from xclim_amd.timeaxis import TimeAxis

    T, cells, nq = 365 * 3, 6, 5
    ta = TimeAxis.daily("2001-01-01", T, "noleap")
    t = np.arange(T)[:, None]
    ref = np.round(20 + 10 * np.sin(2 * np.pi * t / 365) + rng.normal(0, 3, (T, cells)), 1).astype(np.float32)
    hist = (ref[::-1] * 1.01 + rng.normal(0, 1, (T, cells))).astype(np.float32)
    ref[rng.random(ref.shape) < 0.04] = np.nan
    hist[:, 3] = np.nan
    ref[2:5, 1] = np.inf
    hist[7, 1] = -np.inf
    ref[:, 4] = -ref[:, 4]
    hist[:, 5] = 0.0
    rows0, enter, leave = xsdba.Grouper("time.dayofyear", 7).ring_schedule(ta)
    G = 16
    q = (np.arange(nq) + 0.5) / nq
    d_ref, d_hist = sim.to_device(ref), sim.to_device(hist)
    af, hq, sc, muh = (a.get() for a in K.eqm_train_window(sim, d_ref, d_hist, rows0, enter[:G - 1], leave[:G - 1], q, kind, normalised=True))
    inv = "-" if kind == "+" else "/"
    rows = rows0.copy()
    exact = 0
    for g in range(G):
        rg, hg = K.select_rows(sim, d_ref, rows), K.select_rows(sim, d_hist, rows)
        mu_r, _ = K.poly_trend(sim, rg, 0)
        mu_h, _ = K.poly_trend(sim, hg, 0)
        a_g, h_g = K.eqm_train(sim, K.trend_apply(sim, rg, mu_r, None, inv), K.trend_apply(sim, hg, mu_h, None, inv), q, kind)
        mr, mh = mu_r.get(), mu_h.get()
        np.testing.assert_allclose(muh[g], mh, rtol=1e-14, err_msg=f"group {g}")
        with np.errstate(all="ignore"):
            np.testing.assert_allclose(sc[g], mr - mh if kind == "+" else mr / mh, rtol=1e-14, err_msg=f"group {g}")
        same = (muh[g] == mh) | (np.isnan(mh) & np.isnan(muh[g]))
        exact += int(same.sum())
        np.testing.assert_array_equal(hq[g][:, same], h_g.get()[:, same], err_msg=f"group {g}")
        np.testing.assert_allclose(hq[g], h_g.get(), rtol=1e-6, atol=1e-6, err_msg=f"group {g}")
        np.testing.assert_allclose(af[g], a_g.get(), rtol=1e-6, atol=2e-6, err_msg=f"group {g}")
        if g + 1 < G:
            for y in range(enter.shape[1]):
                hit = np.nonzero(rows == leave[g, y])[0]
                rows[hit[0] if leave[g, y] >= 0 and len(hit) else np.nonzero(rows < 0)[0][0]] = enter[g, y]
    assert exact > G * cells // 2
    assert np.isnan(hq[:, :, 3]).all() and np.isfinite(hq[:, :, 0]).all()
    if kind == "*":
        assert np.isnan(hq[:, :, 5]).all()         # 0 / 0: no sample survives the normalisation
        assert (np.diff(af[:, :, 4], axis=1) * 0 == 0).all()


@pytest.mark.parametrize("interp", ["nearest", "linear"])
def test_qdm_small_groups_on_the_simulation(sim, rng, interp):
    """k_qdm_groups (qdm.hip, round 6: all groups of a day-of-year grouping ranked in registers in one launch) against xh_qdm_adjust
    on every group's gathered rows, bit for bit: 12 groups of 0 to 9 rows, NaN samples, ties, the two zeros, a constant cell, NaN
    factors.  30 to 64 rows per group: tests/test_gpu_api.py::test_qdm_small_groups_match_per_group_calls on the GPU."""
    T, C, G, nq = 60, 6, 12, 7
    x = np.round(rng.normal(3, 2, (T, C)), 1).astype(np.float32)
    x[rng.random(x.shape) < 0.1] = np.nan
    x[:, 1] = np.where(rng.random(T) < 0.5, 0.0, x[:, 1])
    x[::3, 1] = -0.0
    x[:, 2] = 4.0
    gid = rng.integers(0, G, T)
    gid[gid == 7] = 8
    perm = np.argsort(gid, kind="stable")
    offs = np.concatenate([[0], np.cumsum(np.bincount(gid, minlength=G))])
    af = rng.normal(1, 0.3, (G, nq, C)).astype(np.float32)
    af[:, 2:4, 3] = np.nan
    af[:, 1:, 4] = np.nan
    q = (np.arange(nq) + 0.5) / nq
    d_x, d_af = sim.to_device(x), sim.to_device(af)
    for kind, extrap in (("+", "constant"), ("*", "nan"), ("factor", "constant")):
        out = sim.to_device(np.full((T, C), -3.0, np.float32))
        got = K.qdm_adjust_groups(sim, d_x, perm, offs, d_af, q, kind, interp, extrap, out=out).get()
        for g in range(G):
            rows = perm[offs[g]:offs[g + 1]]
            if len(rows):
                exp = K.qdm_adjust(sim, K.select_rows(sim, d_x, rows), sim.to_device(af[g]), q, kind, interp, extrap).get()
                np.testing.assert_array_equal(got[rows], exp, err_msg=f"{kind} group {g}")
    # a group of more than 64 rows is not this kernel's: the caller gathers it
    assert K.qdm_adjust_groups(sim, d_x, np.r_[np.arange(T), np.arange(10)], np.array([0, T + 10]), d_af, q) is None


@pytest.mark.parametrize("kind", ["+", "*"])
def test_small_group_training_on_the_simulation(sim, rng, kind):
    """k_group_quantiles (winsel.hip, round 6: the training of all groups of a day-of-year grouping without a window in one launch)
    against xh_eqm_train — and, normalised, xh_poly_trend + xh_trend_apply + xh_eqm_train — on every group's gathered rows, bit
    for bit (tables, means, scaling): 10 groups of 0 to 9 rows, NaN samples, ties, an infinity, a negative-mean cell, zeros."""
    T, C, G, nq = 50, 7, 10, 5
    x = np.round(rng.normal(6, 3, (T, C)), 1).astype(np.float32)
    y = (x[::-1] * 1.1 + rng.normal(0, 1, (T, C))).astype(np.float32)
    x[rng.random(x.shape) < 0.1] = np.nan
    y[:, 2] = np.nan
    x[3, 1] = np.inf
    x[:, 4] = -x[:, 4]
    y[:, 5] = 0.0
    gid = rng.integers(0, G, T)
    gid[gid == 6] = 7
    perm = np.argsort(gid, kind="stable")
    offs = np.concatenate([[0], np.cumsum(np.bincount(gid, minlength=G))])
    q = (np.arange(nq) + 0.5) / nq
    d_x, d_y = sim.to_device(x), sim.to_device(y)
    af, hq = (a.get() for a in K.eqm_train_groups(sim, d_x, d_y, perm, offs, q, kind))
    naf, nhq, sc, muh = (a.get() for a in K.eqm_train_groups(sim, d_x, d_y, perm, offs, q, kind, normalised=True))
    inv = "-" if kind == "+" else "/"
    for g in range(G):
        rows = perm[offs[g]:offs[g + 1]]
        if not len(rows):
            assert np.isnan(hq[g]).all() and np.isnan(nhq[g]).all() and np.isnan(muh[g]).all()
            continue
        xg, yg = K.select_rows(sim, d_x, rows), K.select_rows(sim, d_y, rows)
        a_g, h_g = K.eqm_train(sim, xg, yg, q, kind)
        np.testing.assert_array_equal(hq[g], h_g.get(), err_msg=f"group {g}")
        np.testing.assert_array_equal(af[g], a_g.get(), err_msg=f"group {g}")
        mu_x, _ = K.poly_trend(sim, xg, 0)
        mu_y, _ = K.poly_trend(sim, yg, 0)
        a_g, h_g = K.eqm_train(sim, K.trend_apply(sim, xg, mu_x, None, inv), K.trend_apply(sim, yg, mu_y, None, inv), q, kind)
        np.testing.assert_array_equal(muh[g], mu_y.get(), err_msg=f"group {g}")
        with np.errstate(all="ignore"):
            np.testing.assert_array_equal(sc[g], mu_x.get() - mu_y.get() if kind == "+" else mu_x.get() / mu_y.get(), err_msg=f"group {g}")
        np.testing.assert_array_equal(nhq[g], h_g.get(), err_msg=f"group {g}")
        np.testing.assert_array_equal(naf[g], a_g.get(), err_msg=f"group {g}")
    assert K.eqm_train_groups(sim, d_x, d_y, np.tile(np.arange(T), 2), np.array([0, 2 * T]), q, kind) is None   # 100 rows in a group

