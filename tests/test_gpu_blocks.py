"""Block adapter (xclim_amd/blocks.py): slabs of cells streamed through the device over the copy
lanes must give exactly what one whole-array call gives — every op on the path is independent per grid cell."""

import numpy as np
import pytest

from oracle import indices as oidx
from oracle import sdba as osdba
from oracle.timeutil import OTime
from xclim_amd import kernels as K
from xclim_amd.blocks import map_cell_blocks
from xclim_amd.timeaxis import TimeAxis

pytestmark = pytest.mark.gpu


def _field(rng, T, shape):
    t = np.arange(T).reshape((T,) + (1,) * len(shape))
    x = (288 + 12 * np.sin(2 * np.pi * (t - 100) / 365) + rng.normal(0, 3, (T,) + shape)).astype(np.float32)
    x[rng.random(x.shape) < 0.002] = np.nan
    return x


@pytest.mark.parametrize("pinned", [False, True])
@pytest.mark.parametrize("block_cells", [64, 256, 1000, 4096])
def test_blocks_tx90p_chain(dev, rng, pinned, block_cells):
    """percentile_doy -> threshold_count per slab; two outputs of different dtype and row count; ragged last slab and a
    cell count that is not a multiple of 4 (1003 = 17 x 59)."""
    T, shape = 730, (17, 59)
    host = _field(rng, T, shape)
    if pinned:
        x = dev.pinned_empty(host.shape, np.float32)
        x[...] = host
    else:
        x = host
    ta, ot = TimeAxis.daily("2001-01-01", T), OTime.standard("2001-01-01", T)
    tb, years, doys = ta.doy_table()
    seg, _ = ta.segments("YS")
    tidx = np.searchsorted(doys, ta.doy).astype(np.int32)

    def chain(d, xs):
        p = K.percentile_doy(d, xs, tb, 5, [90.0])
        table = p.reshape(len(doys), xs.shape[1])
        cnt, _ = K.threshold_count(d, xs, ">", seg, doy_table=table, tidx=tidx, want_valid=False)
        return cnt, table

    cnt, table = map_cell_blocks(chain, [x], block_cells=block_cells, device=dev, pinned_out=pinned)
    whole_cnt, whole_table = chain(dev, dev.to_device(host.reshape(T, -1)))
    assert cnt.shape == (2,) + shape and table.shape == (len(doys),) + shape and table.dtype == np.float64
    np.testing.assert_array_equal(cnt.reshape(2, -1), whole_cnt.get())
    np.testing.assert_array_equal(table.reshape(len(doys), -1), whole_table.get())
    from oracle import calendar as ocal

    exp, edoys = ocal.percentile_doy(host, ot, 5, 90.0)
    np.testing.assert_array_equal(cnt, oidx.tx90p(host, exp[..., 0], edoys, ot, "YS", ">"))


def test_blocks_eqm_three_inputs(dev, rng):
    """ref / hist / sim with different lengths; a (T, slab) output; single-output functions return one array."""
    shape = (9, 31)
    ref = rng.gamma(2.0, 2.0, (400,) + shape).astype(np.float32)
    hist = (rng.gamma(2.0, 2.5, (400,) + shape) + 0.5).astype(np.float32)
    sim = (rng.gamma(2.0, 2.5, (531,) + shape) + 0.7).astype(np.float32)
    sim[rng.random(sim.shape) < 0.01] = np.nan
    q = (np.arange(20) + 0.5) / 20

    def eqm(d, r, h, s):
        af, hq = K.eqm_train(d, r, h, q, "+")
        return K.eqm_adjust(d, s, af, hq, "+", "linear", "constant")

    scen = map_cell_blocks(eqm, [ref, hist, sim], block_cells=100, device=dev, pinned_out=False)
    assert isinstance(scen, np.ndarray) and scen.shape == sim.shape
    af, hq = osdba.eqm_train(ref, hist, 20, "+")
    exp = osdba.eqm_adjust(sim, af, hq, "+", "linear", "constant")
    np.testing.assert_allclose(scen, exp, rtol=1e-6, equal_nan=True)
    one = map_cell_blocks(eqm, [ref, hist, sim], device=dev)  # default slab: everything in one block
    np.testing.assert_array_equal(one, scen)
    pre = dev.pinned_empty(sim.shape, np.float32)  # preallocated (page-locked) output
    res = map_cell_blocks(eqm, [ref, hist, sim], block_cells=64, device=dev, out=pre)
    assert res is pre or res.base is not None
    np.testing.assert_array_equal(pre, scen)
    with pytest.raises(ValueError):
        map_cell_blocks(eqm, [ref, hist, sim], device=dev, out=np.empty(sim.shape, np.float64))


def test_blocks_register_and_errors(dev, rng):
    x = _field(rng, 365, (8, 16))
    seg = np.array([0, 365], dtype=np.int64)
    f = lambda d, xs: K.run_stats(d, xs, "max", 1, seg, fused_op=">", thresh=290.0, want_valid=False)[0]
    ref = map_cell_blocks(f, [x], block_cells=32, device=dev, pinned_out=False)
    dev.register(x)
    try:
        assert dev.is_pinned(x) and dev.is_pinned(x[10:20])
        got = map_cell_blocks(f, [x], block_cells=32, device=dev)
    finally:
        dev.unregister(x)
    assert not dev.is_pinned(x)
    np.testing.assert_array_equal(got, ref)
    from oracle import run_length as orl

    np.testing.assert_array_equal(ref[0], orl.longest_run(x > np.float32(290.0)))
    with pytest.raises(ValueError):
        map_cell_blocks(f, [x, x[:, :4]], device=dev)
    with pytest.raises(ValueError):
        map_cell_blocks(lambda d, xs: K.transpose(d, xs), [x], block_cells=32, device=dev)  # not (R, slab)
    with pytest.raises(ValueError):
        map_cell_blocks(f, [np.asfortranarray(x)], device=dev)


def test_dask_style_map_blocks(dev, rng):
    """xclim_amd.dask_adapter: index functions as map_blocks callees (the role xr.map_blocks / apply_ufunc(dask=
    "parallelized") play in the reference, indices/helpers.py:898-974).  dask is not installed here: the numpy walk over
    a dask-style chunk grid drives the SAME callee that dask.array.map_blocks would call per chunk; ragged chunk edges,
    one- and two-variable indices, percentile-free thresholds; stitched results equal the unchunked call."""
    from xclim_amd import dask_adapter as xda
    from xclim_amd import indices as xi
    from xclim_amd.timeaxis import TimeAxis

    T, Y, X = 730, 13, 21
    ta = TimeAxis.daily("2001-01-01", T, "noleap")
    tas = (285 + 8 * rng.standard_normal((T, Y, X))).astype(np.float32)
    tas[rng.random(tas.shape) < 0.002] = np.nan
    tasmax = tas + np.abs(rng.normal(3, 1, tas.shape)).astype(np.float32)
    for name, args, kw, arrs in (("tx_days_above", (), dict(thresh=290.0, freq="MS"), [tasmax]),
                                 ("tg_mean", (), dict(freq="YS"), [tas]),
                                 ("maximum_consecutive_tx_days", (), dict(thresh=288.0, freq="YS"), [tasmax]),
                                 ("daily_temperature_range", (), dict(freq="QS-DEC"), [tas, tasmax])):
        full = getattr(xi, name)(*arrs, *args, time=ta, device=dev, **kw)
        for chunks in ((5, 8), (13, 21), ((4, 9), (10, 10, 1))):
            got = xda.map_blocks(name, arrs, ta, *args, chunks=chunks, device=dev, **kw)
            np.testing.assert_array_equal(got, full, err_msg=f"{name} chunks={chunks}")
    grid = xda.chunk_grid((13, 21), (5, 8))
    assert len(grid) == 3 * 3 and grid[-1] == (slice(10, 13), slice(16, 21))
    f = xda.block_function("tg_mean", ta, freq="YS", device=dev)
    with pytest.raises(ValueError, match="whole time axis"):
        f(tas[:100])
    with pytest.raises(ValueError, match="expected 1 data block"):
        f(tas, tas)
