"""Edge cases through the C ABI: empty grids, single cells, one time step, ragged (non multiple-of-4, misaligned) cell
counts, all-NaN fields, error codes.  The reference tests the same corners with xarray (empty arrays, all-NaN slices:
tests/test_run_length.py:100-130, tests/test_utils.py:28-73)."""
import numpy as np
import pytest

from oracle import generic as ogen
from oracle import run_length as orl
from oracle import sdba as osdba
from oracle.timeutil import OTime
from xclim_amd import generic as xgen
from xclim_amd import kernels as K
from xclim_amd import run_length as xrl
from xclim_amd._capi import XclimHipError
from xclim_amd.timeaxis import TimeAxis

pytestmark = pytest.mark.gpu


def _axes(T):
    return TimeAxis.daily("2001-01-01", T, "standard"), OTime.standard("2001-01-01", T)


@pytest.mark.parametrize("cells", [(0,), (1,), (3,), (5, 1), (257,), (1021,)])
def test_ragged_and_empty_grids(dev, rng, cells):
    T = 400
    ta, ot = _axes(T)
    x = (rng.normal(285, 6, (T,) + cells)).astype(np.float32)
    if x.size:
        x[rng.random(x.shape) < 0.02] = np.nan
    np.testing.assert_array_equal(xgen.threshold_count(x, ">", 288.0, ta, "MS", device=dev), ogen.threshold_count(x, ">", 288.0, ot, "MS"))
    got = xgen.select_resample_op(x, "mean", ta, "YS", device=dev)
    assert got.shape == (2,) + cells
    if x.size:
        np.testing.assert_allclose(got, ogen.select_resample_op(x, "mean", ot, "YS"), rtol=1e-6, equal_nan=True)
    m = x > 287.0
    np.testing.assert_array_equal(xrl.rle_statistics(m, "max", 1, freq="YS", time=ta, device=dev),
                                  orl.rle_statistics(m, "max", 1, time=ot, freq="YS"))
    assert xrl.rle(m, device=dev).shape == m.shape
    q = osdba.equally_spaced_nodes(10)
    out = K.quantile_series(dev, dev.to_device(x.reshape(T, -1)), q).get()
    assert out.shape == (10, int(np.prod(cells)))
    if x.size:
        np.testing.assert_allclose(out, osdba.quantile(x.reshape(T, -1), q), rtol=1e-6, equal_nan=True)


def test_single_step_and_all_nan(dev):
    ta, ot = _axes(1)
    x = np.array([[290.0, np.nan, 280.0]], np.float32)
    np.testing.assert_array_equal(xgen.threshold_count(x, ">", 285.0, ta, "YS", device=dev), [[1, 0, 0]])
    np.testing.assert_array_equal(xrl.rle_statistics(x > 285.0, "max", 1, device=dev), [1, 0, 0])
    q = np.array([0.1, 0.5, 0.9])
    out = K.quantile_series(dev, dev.to_device(x), q).get()
    np.testing.assert_allclose(out, osdba.quantile(x, q), equal_nan=True)
    T = 40
    ta, ot = _axes(T)
    nan = np.full((T, 6), np.nan, np.float32)
    np.testing.assert_array_equal(xgen.threshold_count(nan, ">", 0.0, ta, "YS", device=dev), np.zeros((1, 6)))
    assert np.isnan(xgen.select_resample_op(nan, "max", ta, "YS", device=dev)).all()
    # all-NaN: 0 on the N-D path (tests/test_run_length.py:89-91); the 1-D ufunc path that a 6-cell grid takes by default
    # ends in np.nanmax of an empty selection -> NaN (statistics_run_1d, rl:1418-1437)
    np.testing.assert_array_equal(xrl.rle_statistics(nan, "max", 1, device=dev, ufunc_1dim=False), np.zeros(6))
    assert np.isnan(xrl.rle_statistics(nan, "max", 1, device=dev)).all()
    np.testing.assert_array_equal(xrl.windowed_run_count(nan, 1, device=dev), np.zeros(6))
    assert np.isnan(K.quantile_series(dev, dev.to_device(nan), q).get()).all()


def test_misaligned_views_take_the_scalar_path(dev, rng):
    """A view that starts 4 bytes into an allocation is not 16-byte aligned: the kernels must fall back to VEC = 1."""
    T, C = 120, 64
    big = rng.normal(0, 1, (T, C + 4)).astype(np.float32)
    d = dev.to_device(big.reshape(-1))
    ta, ot = _axes(T)
    seg, _ = ta.segments("MS")
    # the (T, C) field with row stride C + 4 starting at element 1
    view = dev.wrap(d.ptr + 4, (T, C + 4), np.float32)
    flat = np.concatenate([big.reshape(-1)[1:], [0.0]]).astype(np.float32).reshape(T, C + 4)
    cnt, val = K.threshold_count(dev, view, ">", seg, scalar=0.25)
    np.testing.assert_array_equal(cnt.get()[:, :C], ogen.threshold_count(flat[:, :C], ">", 0.25, ot, "MS"))


def test_error_codes(dev):
    x = dev.to_device(np.zeros((10, 4), np.float32))
    with pytest.raises(ValueError):
        K.threshold_count(dev, x, "=>", np.array([0, 10], np.int64), scalar=0.0)
    with pytest.raises(XclimHipError):
        K.threshold_count(dev, x, ">", np.array([0, 11], np.int64), scalar=0.0)  # segment past the end
    with pytest.raises(XclimHipError):
        K.run_stats(dev, x, "max", 0, np.array([0, 10], np.int64))  # window < 1
    with pytest.raises(XclimHipError):
        K.quantile_series(dev, x, np.linspace(0.01, 0.99, 65))  # more than 64 quantiles
