"""Pin the oracle: (1) against golden vectors produced by EXECUTING the reference's own pure-numpy/njit function
bodies (tests/golden/make_golden.py -> reference_vectors.npz), (2) against the known answers of the reference's
tests for the quantile code (tests/test_utils.py:28-73 of the reference)."""

import os
import warnings

import numpy as np
import pytest

from oracle import quantile as oq
from oracle import run_length as orl

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.npz"))


@pytest.mark.parametrize("k", range(7))
@pytest.mark.parametrize("typ,ab", [("t7", (1.0, 1.0)), ("t8", (1.0 / 3.0, 1.0 / 3.0))])
def test_calc_perc_matches_reference_bitwise(k, typ, ab):
    x = G[f"q_in_{k}"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = oq.calc_perc(x, list(G["q_pers"]), *ab)
    ref = G[f"q_{typ}_{k}"]
    # fp32 in -> fp64 out, except the length-1 axis shortcut which returns the value itself (utl:508-510)
    assert got.dtype == ref.dtype == (np.float32 if x.shape[-1] == 1 else np.float64)
    np.testing.assert_array_equal(got, ref)


def test_calc_perc_float64_input():
    np.testing.assert_array_equal(oq.calc_perc(G["q64_in"], list(G["q_pers"]), 1 / 3, 1 / 3), G["q64_t8"])


def test_reference_known_answers():
    arr = np.asarray([15.0, 20.0, 35.0, 40.0, 50.0])
    # R quantile(type=7) -> 29 ; type=8 -> 27 (reference tests/test_utils.py:29-55)
    assert oq.nan_calc_percentiles(arr, [40.0], alpha=1, beta=1)[()] == 29 == G["ka_type7"][()]
    res = oq.nan_calc_percentiles(np.stack([arr, arr]), [40.0], alpha=1 / 3.0, beta=1 / 3.0)
    assert np.all(res[0] == 27) and np.all(G["ka_type8"] == 27)
    res2d = oq.nan_calc_percentiles(np.stack([arr, arr]), [40.0])
    assert np.all(res2d[0] == 29)
    # all-NaN -> NaN ; empty -> NaN ; partial NaN type 8 -> 42.0 (tests/test_utils.py:57-73)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert np.isnan(oq.nan_calc_percentiles(np.asarray([np.nan]), [50.0]))
    assert np.isnan(oq.nan_calc_percentiles(np.asarray([])))
    assert oq.nan_calc_percentiles(np.asarray([np.nan, 41.0, 41.0, 43.0, 43.0]), [50.0], alpha=1 / 3.0, beta=1 / 3.0)[()] == 42.0
    assert G["ka_partial_nan"][()] == 42.0


def test_against_numpy_quantile_methods():
    rng = np.random.default_rng(3)
    x = rng.normal(size=(200, 31)).astype(np.float32)
    q = np.array([0.05, 0.5, 0.9])
    got7 = oq.nan_quantile(x, q, axis=1, alpha=1, beta=1)
    np.testing.assert_allclose(got7, np.quantile(x.astype(np.float64), q, axis=1), rtol=3e-7)
    got8 = oq.nan_quantile(x, q, axis=1, alpha=1 / 3, beta=1 / 3)
    np.testing.assert_allclose(got8, np.quantile(x.astype(np.float64), q, axis=1, method="median_unbiased"), rtol=3e-7)


@pytest.mark.parametrize("index", ["last", "first"])
def test_cumsum_reset_matches_reference_njit_body(index):
    b = G["cs_in"]  # (6, 7, 50) uint8, core dim last as apply_ufunc hands it
    got = np.moveaxis(orl.cumsum_reset(np.moveaxis(b, -1, 0).astype(bool), index), 0, -1)
    np.testing.assert_array_equal(got, G[f"cs_{index}"])
    if index == "last":
        gotf = np.moveaxis(orl.cumsum_reset(np.moveaxis(b, -1, 0).astype(np.float32), index), 0, -1)
        assert gotf.dtype == np.float32
        np.testing.assert_array_equal(gotf, G["cs_last_f32"])


def test_rle_1d_matches_reference_njit_body():
    v, l, p = orl.rle_1d(G["rle1d_in"])
    np.testing.assert_array_equal(v, G["rle1d_v"])
    np.testing.assert_array_equal(l, G["rle1d_l"])
    np.testing.assert_array_equal(p, G["rle1d_p"])
    # docstring example of the reference (rl:1368-1370)
    v, l, p = orl.rle_1d([0, 0, 1, 1, 1, 0, 0, 1, 1, 1, 1])  # fmt: skip
    np.testing.assert_array_equal(l, [2, 3, 2, 4])
    np.testing.assert_array_equal(p, [0, 2, 5, 7])


def test_numba_quantile_mode_stays_within_the_bar():
    """xsdba's jitted ``nbutils._quantile`` runs numba's np.nanquantile, whose lerp is
    ``lower (1 - m) + upper m`` on ``rank = 1 + (n - 1) q`` — not numpy's ``_lerp`` that the oracle's contract mode (and the
    kernels, bit for bit) follow.  Same order statistics, other rounding: on fields of the fixtures' size
    (tests/golden/make_sdba_golden.py: 4 years x 3 x 4, temperature in K and precipitation in mm/d, 0.2 % NaN) the two
    modes differ by at most a few float32 ulps — documented bound 2e-7 relative, a fifth of north_star's 1e-6."""
    from oracle import sdba as osdba

    rng = np.random.default_rng(20260926)
    T, Y, X = 365 * 4, 3, 4
    t = np.arange(T)[:, None, None]
    temp = (288 + 12 * np.sin(2 * np.pi * (t - 100) / 365) + rng.normal(0, 3.0, (T, Y, X))).astype(np.float32)
    temp[rng.random(temp.shape) < 0.002] = np.nan
    pr = np.where(rng.random((T, Y, X)) < 0.3, rng.gamma(0.8, 8.0, (T, Y, X)), 0.0).astype(np.float32) + np.float32(1e-3)
    worst = 0.0
    for x in (temp, pr, temp[:365], pr[:31]):
        for q in (osdba.equally_spaced_nodes(20), osdba.equally_spaced_nodes(15, eps=1e-6), np.array([0.0, 0.5, 1.0])):
            a, b = osdba.quantile(x, q), osdba.quantile(x, q, mode="numba")
            assert a.dtype == b.dtype == np.float32 and a.shape == b.shape
            rel = np.abs(a.astype(np.float64) - b) / np.maximum(np.abs(a), 1e-30)
            worst = max(worst, float(rel.max()))
    assert 0.0 < worst < 2e-7, worst   # (they DO differ: the bitwise GPU == oracle claim says nothing about upstream's last ulp)
