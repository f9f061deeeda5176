"""Tier 2 of the drop-in boundary, without xarray: the numpy callees that xarray's apply_ufunc would call,
invoked exactly the way apply_ufunc invokes them — core dim LAST as a non-contiguous view, leading dims arbitrary.

* ``_cumsum_reset_np(arr, index, one)`` (reference indices/run_length.py:143-151, call site :209-216): mutates and returns
  ``arr``; pinned to the reference's own function body, executed from /root/reference when the golden vectors were made
  (tests/golden/make_golden.py) and restated in oracle/run_length.py.
* ``calc_perc(arr, percentiles, alpha, beta, copy)`` (core/utils.py:279-323, call site core/calendar.py:469-479).
"""
import numpy as np
import pytest

from oracle import quantile as oq
from oracle import run_length as orl
from xclim_amd import patch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32, np.float64])
@pytest.mark.parametrize("index", ["last", "first"])
def test_cumsum_reset_np_on_the_apply_ufunc_view(dev, rng, dtype, index):
    T, Y, X = 97, 5, 7
    base = (rng.random((T, Y, X)) < 0.6).astype(dtype)          # what the DataArray holds: (time, lat, lon), C order
    arr = np.moveaxis(base, 0, -1)                              # apply_ufunc: core dim moved LAST -> a strided view
    assert not arr.flags.c_contiguous and arr.base is base
    want = np.moveaxis(orl.cumsum_reset_np(base.copy(), index, dtype(1)), 0, -1)
    out = patch.cumsum_reset_np(arr, index, dtype(1), device=dev)
    assert out is arr                                            # contract: mutate and return the argument
    assert out.dtype == dtype
    np.testing.assert_array_equal(out, want)
    np.testing.assert_array_equal(base, np.moveaxis(want, -1, 0))  # ... through the view into the caller's array


def test_cumsum_reset_np_reference_docstring_example(dev):
    """100110111 -> 100120123 (index "last", rl:146) and -> 100210321 (index "first")."""
    a = np.array([[1, 0, 0, 1, 1, 0, 1, 1, 1]], dtype=np.uint8)
    np.testing.assert_array_equal(patch.cumsum_reset_np(a.copy(), "last", np.uint8(1), device=dev), [[1, 0, 0, 1, 2, 0, 1, 2, 3]])
    np.testing.assert_array_equal(patch.cumsum_reset_np(a.copy(), "first", np.uint8(1), device=dev), [[1, 0, 0, 2, 1, 0, 3, 2, 1]])
    one_d = np.array([1, 1, 0, 1], dtype=np.float32)            # vectorised call on a single series
    np.testing.assert_array_equal(patch.cumsum_reset_np(one_d, "last", np.float32(1), device=dev), [1, 2, 0, 1])
    empty = np.zeros((3, 0), np.uint8)
    assert patch.cumsum_reset_np(empty, "last", np.uint8(1), device=dev) is empty
    with pytest.raises(ValueError):
        patch.cumsum_reset_np(a, "middle", np.uint8(1), device=dev)


@pytest.mark.parametrize("alpha,beta", [(1.0, 1.0), (1.0 / 3.0, 1.0 / 3.0)])
def test_calc_perc_on_the_stack_dim_view(dev, rng, alpha, beta):
    """percentile_doy stacks (year, window) into `stack_dim` and apply_ufunc moves it last: (dayofyear, lat, lon, stack_dim)
    arrives as a view of a differently ordered array.  Result: (dayofyear, lat, lon, percentiles) float64."""
    D, Y, X, N = 11, 3, 4, 45
    store = rng.normal(280, 5, (N, D, Y, X)).astype(np.float32)
    store[rng.random(store.shape) < 0.05] = np.nan
    store[:, 2, 1, 1] = np.nan                                    # an all-NaN slice -> NaN
    arr = np.moveaxis(store, 0, -1)
    assert not arr.flags.c_contiguous
    keep = arr.copy()
    pers = [10.0, 50.0, 90.0]
    got = patch.calc_perc(arr, percentiles=pers, alpha=alpha, beta=beta, copy=True, device=dev)
    assert got.shape == (D, Y, X, 3) and got.dtype == np.float64
    want = np.moveaxis(oq.nan_quantile(keep, np.array(pers) / 100.0, axis=-1, alpha=alpha, beta=beta), 0, -1)
    np.testing.assert_allclose(got, want, rtol=1e-12, equal_nan=True)
    np.testing.assert_array_equal(arr, keep)                     # the input is never modified
    med = patch.calc_perc(arr, device=dev)                       # percentiles=None -> the median only, alpha = beta = 1
    want_med = oq.nan_quantile(keep, np.array([0.5]), axis=-1, alpha=1.0, beta=1.0)[0]
    assert med.shape == (D, Y, X, 1)
    np.testing.assert_allclose(med[..., 0], want_med, rtol=1e-12, equal_nan=True)
