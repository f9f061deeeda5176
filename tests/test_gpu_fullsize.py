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
