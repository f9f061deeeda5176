"""BASELINE.json full sizes on one MI355X, checked through size-independent properties + sampled oracle parity.

Inputs are generated on the device by the counter-based generator (xh_fill_synthetic); oracle/synth.py restates it
bit-exactly on the host, so any set of cells can be recomputed by the oracle without moving the 1.5 - 15 GB fields.
Properties used: partition (monthly counts sum to the annual count), monotonicity in the percentile, fused == unfused,
sub-period bounds of run lengths, idempotence of the quantile mapping (ref == hist -> scen == sim bit for bit),
sortedness of the quantile nodes, and a checksum of the sampled cells against the oracle.
"""
import numpy as np
import pytest

from oracle import calendar as ocal
from oracle import indices as oidx
from oracle import sdba as osdba
from oracle import synth
from oracle.timeutil import OTime
from xclim_amd import kernels as K
from xclim_amd.timeaxis import TimeAxis

pytestmark = pytest.mark.gpu

Y, X = 1440, 720
C = Y * X


def _sample_cells(rng, n):
    return np.sort(rng.choice(C, size=n, replace=False))


def test_tx90p_full_size(dev, rng):
    """configs[1]: tx90p on 365 x 1440 x 720 fp32 (percentile_doy window 5, per 90 -> threshold_count)."""
    T = 365
    ta, ot = TimeAxis.daily("2001-01-01", T, "noleap"), OTime.noleap(2001, T)
    tb, years, doys = ta.doy_table()
    base = synth.seasonal_base(T)
    x = K.fill_synthetic(dev, T, C, 0, 2, base, 3.0, nan_per_million=200)
    seg_y, _ = ta.segments("YS")
    seg_m, _ = ta.segments("MS")
    tidx = np.searchsorted(doys, ta.doy).astype(np.int32)
    counts = {}
    for per in (10.0, 50.0, 90.0):
        p = K.percentile_doy(dev, x, tb, 5, [per])
        table = p.reshape(len(doys), C)
        cy, vy = K.threshold_count(dev, x, ">", seg_y, doy_table=table, tidx=tidx)
        cm, vm = K.threshold_count(dev, x, ">", seg_m, doy_table=table, tidx=tidx)
        cy_h, cm_h = cy.get(), cm.get()
        # partition: the 12 monthly counts add up to the annual count, for every cell; same for the valid counts
        np.testing.assert_array_equal(cm_h.sum(axis=0), cy_h[0])
        np.testing.assert_array_equal(vm.get().sum(axis=0), vy.get()[0])
        counts[per] = cy_h[0]
        # fused kernel == two-step chain on the whole grid
        period = (np.searchsorted(seg_m, tb[0], side="right") - 1).astype(np.int32)
        fused = K.percentile_doy_count(dev, x, tb, 5, per, ">", period, len(seg_m) - 1)
        assert fused is not None
        np.testing.assert_array_equal(fused[0].get(), cm_h)
        np.testing.assert_array_equal(fused[1].get(), vm.get())
        if per == 50.0:
            # sampled oracle parity (bit-exact integer counts) on cells regenerated on the host
            cells = _sample_cells(rng, 1536)
            xs = synth.fill_synthetic(T, cells, 0, 2, base, 3.0, nan_per_million=200)
            exp, d2 = ocal.percentile_doy(xs, ot, 5, 50.0)
            np.testing.assert_array_equal(cm_h[:, cells], oidx.tx90p(xs, exp[..., 0], d2, ot, "MS"))
            got_tab = np.stack([dev.wrap(table.ptr + int(c) * 8, (1,), np.float64).get() for c in cells[:8]])
            np.testing.assert_allclose(got_tab[:, 0], exp[0, :8, 0], rtol=1e-12)
        del p
    # monotonicity in the percentile: exceedances of a higher percentile can only be fewer
    assert (counts[10.0] >= counts[50.0]).all() and (counts[50.0] >= counts[90.0]).all()
    # with 5 samples the 90th percentile clips to the window maximum (utl:443-452) and the window contains the day
    # itself: no day can exceed it
    assert counts[90.0].max() == 0
    assert 100 < counts[50.0].mean() < 200


def test_cdd_full_size(dev, rng):
    """configs[2]: maximum_consecutive_dry_days on 3650 x 1440 x 720 fp32 (15.1 GB)."""
    T = 3650
    ta, ot = TimeAxis.daily("2001-01-01", T, "noleap"), OTime.noleap(2001, T)
    seg, _ = ta.segments("YS")
    thr = 1.0 / 86400.0
    pr = K.fill_synthetic(dev, T, C, 1, 3, np.zeros(T, np.float32), 40.0 / 86400.0, 0.3)
    per_year, valid = K.run_stats(dev, pr, "max", 1, seg, cut=True, fused_op="<", thresh=thr)
    whole, _ = K.run_stats(dev, pr, "max", 1, np.array([0, T], np.int64), cut=True, fused_op="<", thresh=thr, want_valid=False)
    nocut, _ = K.run_stats(dev, pr, "max", 1, seg, cut=False, fused_op="<", thresh=thr, want_valid=False)
    py, wh, nc = per_year.get(), whole.get()[0], nocut.get()
    # runs cut at the period edges are never longer than the period nor than the longest run of the whole series, and
    # the longest run of the whole series is the longest of the uncut runs (attributed to the period of their first day)
    assert (py <= 365).all() and (py.max(axis=0) <= wh).all()
    np.testing.assert_array_equal(nc.max(axis=0), wh)
    np.testing.assert_array_equal(valid.get(), np.full((10, C), 365, np.int32))
    cells = _sample_cells(rng, 1024)
    xs = synth.fill_synthetic(T, cells, 1, 3, np.zeros(T, np.float32), 40.0 / 86400.0, 0.3)
    np.testing.assert_array_equal(py[:, cells], oidx.maximum_consecutive_dry_days(xs, thr, ot, "YS"))
    np.testing.assert_array_equal(nc[:, cells], oidx.maximum_consecutive_dry_days(xs, thr, ot, "YS", resample_before_rl=False))


def test_eqm_full_size(dev, rng):
    """configs[3] on one year (365 x 1440 x 720; the 30-year size is timed by tools/bench_configs.py): train + adjust."""
    T = 365
    base = synth.seasonal_base(T)
    ref = K.fill_synthetic(dev, T, C, 0, 4, base, 3.0)
    hist = K.fill_synthetic(dev, T, C, 0, 5, base + np.float32(1.5), 3.3)
    sim = K.fill_synthetic(dev, T, C, 0, 6, base + np.float32(3.5), 3.3)
    q = (np.arange(20) + 0.5) / 20
    # idempotence: mapping a distribution onto itself is the identity, bit for bit ("+": af == 0, "*": af == 1)
    af0, hq0 = K.eqm_train(dev, hist, hist, q, "+")
    assert not af0.get().any()
    scen0 = K.eqm_adjust(dev, sim, af0, hq0, "+", "nearest", "constant")
    h = np.zeros(4, np.uint64)
    for i, arr in enumerate((scen0, sim)):
        a = arr.get().view(np.uint32)
        h[2 * i], h[2 * i + 1] = a.sum(dtype=np.uint64), np.bitwise_xor.reduce(a.ravel())
        del a
    assert h[0] == h[2] and h[1] == h[3]  # checksum + xor of the raw bits of scen and sim
    del scen0
    af, hq = K.eqm_train(dev, ref, hist, q, "+")
    hq_h, af_h = hq.get(), af.get()
    assert (np.diff(hq_h, axis=0) >= 0).all()  # quantile nodes are sorted in every cell
    scen = K.eqm_adjust(dev, sim, af, hq, "+", "nearest", "constant")
    cells = _sample_cells(rng, 768)
    refs = synth.fill_synthetic(T, cells, 0, 4, base, 3.0)
    hists = synth.fill_synthetic(T, cells, 0, 5, base + np.float32(1.5), 3.3)
    sims = synth.fill_synthetic(T, cells, 0, 6, base + np.float32(3.5), 3.3)
    oaf, ohq = osdba.eqm_train(refs, hists, q, "+")
    np.testing.assert_allclose(hq_h[:, cells], ohq, rtol=1e-6)
    np.testing.assert_allclose(af_h[:, cells], oaf, rtol=1e-6, atol=1e-5)
    got = np.stack([dev.wrap(scen.ptr + int(c) * 4, (1,), np.float32).get()[0] for c in cells[:16]])  # step 0 of 16 cells
    exp = osdba.eqm_adjust(sims[:, :16], af_h[:, cells[:16]], hq_h[:, cells[:16]], "+", "nearest", "constant")
    np.testing.assert_allclose(got, exp[0], rtol=1e-6)
    # linear interpolation on the full grid stays between the extreme adjustment factors
    scen_l = K.eqm_adjust(dev, sim, af, hq, "+", "linear", "constant")
    d = scen_l.get()
    d -= sim.get()
    assert (d.min(axis=0) >= af_h.min(axis=0) - 1e-3).all() and (d.max(axis=0) <= af_h.max(axis=0) + 1e-3).all()


# ---- the 30-year configurations at their own size (BASELINE configs 4 and 5) ------------------------------------------
def _column_block(dev, arr, T, ncols_total, c0, w):
    """Rows 0..T-1 of cells c0 .. c0+w-1 of a device (T, C) float32 array -> host (T, w), one strided 2-D copy."""
    host = np.empty((T, w), np.float32)
    dev.copy2d(host.ctypes.data, w * 4, arr.ptr + int(c0) * 4, ncols_total * 4, w * 4, T, "d2h")
    return host


def _tx90p_30yr(dev, rng, ncells, cell0, nsample):
    """tx90p on 30 noleap years: percentile_doy (150 samples per doy: the register top-16 kernel + the per-doy count over
    30 periods), sampled cells recomputed by the oracle, bit-exact counts, non-trivial exceedances."""
    T = 10950
    ta, ot = TimeAxis.daily("1981-01-01", T, "noleap"), OTime.noleap(1981, T)
    tb, years, doys = ta.doy_table()
    base = synth.seasonal_base(T)
    x = K.fill_synthetic(dev, T, ncells, 0, 2, base, 3.0, nan_per_million=50, cell0=cell0)
    seg, _ = ta.segments("YS")
    P = len(seg) - 1
    tidx = np.searchsorted(doys, ta.doy).astype(np.int32)
    p = K.percentile_doy(dev, x, tb, 5, [90.0])
    table = p.reshape(len(doys), ncells)
    cnt, val = K.threshold_count(dev, x, ">", seg, doy_table=table, tidx=tidx)
    cnt_h, val_h = cnt.get(), val.get()
    assert cnt_h.shape == (P, ncells) and P == 30
    # ~10 % of the days exceed the 90th percentile of their own climatology
    assert 30.0 < cnt_h.mean() < 43.0 and cnt_h.min() >= 0 and cnt_h.max() < 120
    assert (val_h <= 365).all() and val_h.mean() > 364.9
    # fused kernel (the table is never written) == two-step chain, every cell, every year
    period = (np.searchsorted(seg, tb, side="right") - 1).astype(np.int32)
    period[tb < 0] = -1
    fused = K.percentile_doy_count(dev, x, tb, 5, 90.0, ">", period, P)
    assert fused is not None
    np.testing.assert_array_equal(fused[0].get(), cnt_h)
    np.testing.assert_array_equal(fused[1].get(), val_h)
    del fused
    cells = np.sort(rng.choice(ncells, size=nsample, replace=False))
    xs = synth.fill_synthetic(T, cells + cell0, 0, 2, base, 3.0, nan_per_million=50)
    exp_p, d2 = ocal.percentile_doy(xs, ot, 5, 90.0)
    exp_c = oidx.tx90p(xs, exp_p[..., 0], d2, ot, "YS")
    np.testing.assert_array_equal(cnt_h[:, cells], exp_c)
    assert exp_c.sum() > 0
    # the percentile table itself on the first sampled cells (fp64, same operation order as utl:486-488)
    got = np.stack([dev.wrap(table.ptr + int(c) * 8, (1,), np.float64).get()[0] for c in cells[:16]])
    np.testing.assert_allclose(got, exp_p[0, :16, 0], rtol=1e-12)
    return x, cnt_h


def test_tx90p_30yr_full_grid(dev, rng):
    """BASELINE configs[3]/[4] tx90p half at its own size: 10950 x 1440 x 720 fp32 (45.4 GB), every workgroup of the
    multi-year kernels (k_pdoy_top16 at grid width, per-doy threshold_count over 30 periods)."""
    x, _ = _tx90p_30yr(dev, rng, C, 0, 512)
    x.free()


def test_percentile_doy_30yr_central_percentiles_full_grid(dev, rng):
    """The 30-year field at its own size through the kernels for percentiles in the middle of the distribution
    (k_pdoy_walk): size-independent properties on every (doy, cell) — p25 <= p50 <= p75, all finite, all inside the
    field's range, the median within a degree of the seasonal mean — and sampled cells against the oracle."""
    T = 10950
    ta, ot = TimeAxis.daily("1981-01-01", T, "noleap"), OTime.noleap(1981, T)
    tb, years, doys = ta.doy_table()
    base = synth.seasonal_base(T)
    x = K.fill_synthetic(dev, T, C, 0, 2, base, 3.0, nan_per_million=50)
    per = [25.0, 50.0, 75.0]
    p = K.percentile_doy(dev, x, tb, 5, per)
    ph = p.get()  # (3, 365, C) fp64: 9 GB on the host
    assert ph.shape == (3, len(doys), C) and np.isfinite(ph).all()
    assert (ph[0] <= ph[1]).all() and (ph[1] <= ph[2]).all()
    clim = base[:365].astype(np.float64)[:, None]
    assert np.abs(ph[1] - clim).max() < 1.5  # 150 samples of noise with amplitude 3 around the seasonal cycle
    assert 1.5 < (ph[2] - ph[0]).mean() < 5.0  # the interquartile range of the synthetic noise (amplitude 3)
    cells = np.sort(rng.choice(C, size=128, replace=False))
    xs = synth.fill_synthetic(T, cells, 0, 2, base, 3.0, nan_per_million=50)
    exp_p, _ = ocal.percentile_doy(xs, ot, 5, per)  # (365, cells, 3)
    np.testing.assert_allclose(ph[:, :, cells], np.moveaxis(exp_p, -1, 0), rtol=1e-12)
    x.free()
    p.free()


def _eqm_30yr(dev, rng, ncells, cell0, nsample):
    T = 10950
    base = synth.seasonal_base(T)
    q = (np.arange(20) + 0.5) / 20
    ref = K.fill_synthetic(dev, T, ncells, 0, 4, base, 3.0, cell0=cell0)
    hist = K.fill_synthetic(dev, T, ncells, 0, 5, base + np.float32(1.5), 3.3, cell0=cell0)
    af, hq = K.eqm_train(dev, ref, hist, q, "+")
    ref.free()
    hq_h, af_h = hq.get(), af.get()
    assert np.isfinite(hq_h).all() and (np.diff(hq_h, axis=0) >= 0).all()
    cells = np.sort(rng.choice(ncells, size=nsample, replace=False))
    refs = synth.fill_synthetic(T, cells + cell0, 0, 4, base, 3.0)
    hists = synth.fill_synthetic(T, cells + cell0, 0, 5, base + np.float32(1.5), 3.3)
    oaf, ohq = osdba.eqm_train(refs, hists, q, "+")
    np.testing.assert_allclose(hq_h[:, cells], ohq, rtol=1e-6)
    np.testing.assert_allclose(af_h[:, cells], oaf, rtol=1e-6, atol=1e-5)
    hist.free()
    sim = K.fill_synthetic(dev, T, ncells, 0, 6, base + np.float32(3.5), 3.3, cell0=cell0)
    scen = K.eqm_adjust(dev, sim, af, hq, "+", "nearest", "constant")
    # 64 adjacent FULL columns of scen (two places of the grid) against the oracle fed with the device's own nodes
    for c0 in (int(cells[nsample // 3]) // 64 * 64, max(ncells - 64, 0)):
        w = min(64, ncells - c0)
        got = _column_block(dev, scen, T, ncells, c0, w)
        sims = synth.fill_synthetic(T, np.arange(c0, c0 + w) + cell0, 0, 6, base + np.float32(3.5), 3.3)
        exp = osdba.eqm_adjust(sims, af_h[:, c0:c0 + w], hq_h[:, c0:c0 + w], "+", "nearest", "constant")
        np.testing.assert_allclose(got, exp, rtol=1e-6)
    sim.free()
    scen.free()


def test_eqm_30yr_full_grid(dev, rng):
    """BASELINE configs[3] at its own size: EQM train + adjust on 10950 x 1440 x 720 (ref, hist, sim, scen: 45.4 GB each;
    at most three of them are resident at once)."""
    _eqm_30yr(dev, rng, C, 0, 512)


def test_config5_slab(dev, rng):
    """BASELINE configs[4], the slab ONE of the 8 GPUs owns: 10950 x 360 x 1440 cells of the 2880 x 1440 grid (rank 5's
    cell range of the global counter-based field), tx90p and EQM, sampled cells against the oracle."""
    ncells = 360 * 1440
    cell0 = 5 * ncells
    x, _ = _tx90p_30yr(dev, rng, ncells, cell0, 256)
    x.free()
    _eqm_30yr(dev, rng, ncells, cell0, 256)
