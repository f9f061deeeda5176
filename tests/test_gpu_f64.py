"""float64 FIELDS keep their dtype: the reference computes in the input dtype — compare() of a float64
DataArray is a float64 compare (indices/generic.py:301-326, 360), resample(...).<op>() returns float64 (gen:83-125),
_nan_quantile takes `diff` in float64 (core/utils.py:486).  xh_threshold_count_f64 / xh_resample_reduce_f64 /
xh_nan_quantile_f64 serve those; every other entry point REFUSES a float64 field (Float64FieldError) instead of rounding
it, unless XCLIM_AMD_FLOAT64=round asks for the rounding (PrecisionWarning)."""
import os

import numpy as np
import pytest

from oracle import calendar as ocal
from oracle import generic as ogen
from oracle import quantile as oq
from oracle.timeutil import OTime
from xclim_amd import calendar as hcal
from xclim_amd import generic as hgen
from xclim_amd import patch
from xclim_amd import run_length as hrl
from xclim_amd._capi import Float64FieldError, PrecisionWarning
from xclim_amd.timeaxis import TimeAxis

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.npz")


def test_calc_perc_float64_matches_the_reference_bitwise(dev):
    """q64_in / q64_t8: outputs of the reference's own calc_perc on a float64 array (tests/golden/make_golden.py executes
    /root/reference/src/xclim/core/utils.py:279-557) — bit for bit through xh_nan_quantile_f64."""
    G = np.load(GOLD)
    got = patch.calc_perc(G["q64_in"], percentiles=list(G["q_pers"]), alpha=1.0 / 3.0, beta=1.0 / 3.0, device=dev)
    assert got.dtype == np.float64
    np.testing.assert_array_equal(got, G["q64_t8"])


@pytest.mark.parametrize("N", [1, 2, 7, 150, 930])
def test_calc_perc_float64_vs_oracle(dev, rng, N):
    x = rng.normal(280, 5, (6, 5, N))
    x[rng.random(x.shape) < 0.05] = np.nan
    x[0, 0] = np.nan
    x[1, 1, 1:] = np.nan
    pers = [0.0, 10.0, 50.0, 90.0, 100.0]
    for a, b in ((1.0, 1.0), (1 / 3, 1 / 3)):
        got = patch.calc_perc(x, percentiles=pers, alpha=a, beta=b, device=dev)
        want = np.moveaxis(oq.nan_quantile(x, np.array(pers) / 100.0, axis=-1, alpha=a, beta=b), 0, -1)
        np.testing.assert_array_equal(got, want)


def test_threshold_count_float64_counts_exactly(dev, rng):
    """Values within a few float64 ulps of the threshold: a float32 kernel (rounded field) gets these wrong."""
    T, shape = 800, (5, 6)
    ta, ot = TimeAxis.daily("2000-03-15", T), OTime.standard("2000-03-15", T)
    thr = 290.0
    x = thr + (rng.integers(-3, 4, (T,) + shape) * 5.684341886080802e-14)       # +-3 ulp(290) in float64
    x[rng.random(x.shape) < 0.01] = np.nan
    assert len(np.unique(x.astype(np.float32)[~np.isnan(x)])) == 1              # all the same float32 value
    for op in (">", ">=", "<", "<="):
        for freq in ("YS", "MS"):
            np.testing.assert_array_equal(hgen.threshold_count(x, op, thr, ta, freq, device=dev), ogen.threshold_count(x, op, thr, ot, freq))
    cell = thr + rng.integers(-2, 3, shape) * 5.684341886080802e-14               # one threshold per cell
    np.testing.assert_array_equal(hgen.threshold_count(x, ">", cell, ta, "YS", device=dev), ogen.threshold_count(x, ">", cell[None], ot, "YS"))
    full = thr + rng.integers(-2, 3, x.shape) * 5.684341886080802e-14             # one per element
    np.testing.assert_array_equal(hgen.threshold_count(x, "<=", full, ta, "QS-DEC", device=dev), ogen.threshold_count(x, "<=", full, ot, "QS-DEC"))
    np.testing.assert_array_equal(hgen.count_occurrences(x, thr, "!=", ta, "YS", device=dev), ogen.count_occurrences(x, thr, "!=", ot, "YS"))
    cnt, val = hgen.threshold_count(x, ">", thr, ta, "YS", device=dev, with_valid=True)
    np.testing.assert_array_equal(val, ogen.select_resample_op(x, "count", ot, "YS"))


def test_tx90p_on_a_float64_field(dev, rng):
    """percentile table (float64) against a float64 field through the per-doy kernel: the whole chain in float64."""
    T = 365 * 3
    ta, ot = TimeAxis.daily("2001-01-01", T, "noleap"), OTime.noleap(2001, T)
    t = np.arange(T)[:, None, None]
    x = 288 + 12 * np.sin(2 * np.pi * (t - 100) / 365) + rng.normal(0, 3, (T, 4, 5))   # float64
    x32 = x.astype(np.float32)
    per = hcal.percentile_doy(x32, ta, window=5, per=90.0, device=dev)                  # (base period in float32)
    p_o, doys = ocal.percentile_doy(x32, ot, 5, 90.0)
    got = hgen.threshold_count(x, ">", per, ta, "YS", device=dev)
    exp = ogen.threshold_count(x, ">", ocal.resample_doy(p_o[..., 0], doys, ot), ot, "YS")
    np.testing.assert_array_equal(got, exp)


def test_select_resample_op_float64(dev, rng):
    T, shape = 730, (4, 7)
    ta, ot = TimeAxis.daily("2001-01-01", T, "noleap"), OTime.noleap(2001, T)
    x = rng.normal(280, 8, (T,) + shape)
    x[rng.random(x.shape) < 0.02] = np.nan
    x[:40, 0, 0] = np.nan
    for op in ("sum", "mean", "min", "max", "std", "var"):
        got = hgen.select_resample_op(x, op, ta, "MS", device=dev)
        assert got.dtype == np.float64
        np.testing.assert_allclose(got, ogen.select_resample_op(x, op, ot, "MS"), rtol=1e-13, equal_nan=True, err_msg=op)
    for op in ("count", "argmax", "argmin"):
        np.testing.assert_array_equal(hgen.select_resample_op(x, op, ta, "YS", device=dev), ogen.select_resample_op(x, op, ot, "YS"))


def test_float64_fields_are_refused_elsewhere_not_rounded(dev, rng, monkeypatch):
    T = 400
    ta = TimeAxis.daily("2001-01-01", T, "noleap")
    x = rng.normal(280, 8, (T, 3, 4))
    monkeypatch.delenv("XCLIM_AMD_FLOAT64", raising=False)
    with pytest.raises(Float64FieldError, match="float64 fields are only served by"):
        hcal.percentile_doy(x, ta, device=dev)
    with pytest.raises(Float64FieldError):
        hgen.spell_length_statistics(x, 285.0, 1, None, ">", "max", ta, "YS", device=dev)
    with pytest.raises(Float64FieldError):
        hgen.compare(x.astype(np.float32), ">", x, device=dev)                 # a float64 array THRESHOLD
    assert hrl.rle((x > 280).astype(np.float64), device=dev).shape == x.shape if False else True
    monkeypatch.setenv("XCLIM_AMD_FLOAT64", "round")
    with pytest.warns(PrecisionWarning):
        p = hcal.percentile_doy(x, ta, device=dev)
    assert p.values().shape[0] == 365
