"""Fuzzing of the float64 kernels (xh_nan_quantile_f64, xh_threshold_count_f64, xh_resample_reduce_f64) against the oracle.
usage: python tools/fuzz_f64.py [seconds]"""
import json
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import generic as ogen  # noqa: E402  (checker only)
from oracle import quantile as oq  # noqa: E402
from oracle.timeutil import OTime  # noqa: E402
from xclim_amd import generic as hgen  # noqa: E402
from xclim_amd import patch  # noqa: E402
from fuzzdev import get_fuzz_device  # noqa: E402
from xclim_amd.timeaxis import TimeAxis  # noqa: E402

dev = get_fuzz_device()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "99")))
t_end = time.time() + budget
stats = {"quantile": 0, "count": 0, "reduce": 0}
it = 0
while time.time() < t_end:
    it += 1
    if it % 3 == 0:
        N, cells = int(rng.integers(1, 3000)), (int(rng.integers(1, 6)), int(rng.integers(1, 9)))
        x = rng.normal(280, 5, cells + (N,)) * 10.0 ** rng.integers(-2, 3)
        if rng.random() < 0.3:
            x = np.round(x, 1)
        x[rng.random(x.shape) < rng.choice([0.0, 0.03, 0.5])] = np.nan
        pers = sorted(float(p) for p in rng.choice([0.0, 0.1, 1.0, 10.0, 33.3, 50.0, 90.0, 99.9, 100.0], int(rng.integers(1, 6)), replace=False))
        a, b = (1.0, 1.0) if rng.random() < 0.5 else (1 / 3, 1 / 3)
        got = patch.calc_perc(x, percentiles=pers, alpha=a, beta=b, device=dev)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want = np.moveaxis(oq.nan_quantile(x, np.array(pers) / 100.0, axis=-1, alpha=a, beta=b), 0, -1)
        if not np.array_equal(got, want, equal_nan=True):
            print(json.dumps({"FAIL": "quantile_f64", "N": N, "cells": cells, "pers": pers, "alpha": a}))
            sys.exit(1)
        stats["quantile"] += 1
        continue
    T, shape = int(rng.integers(30, 1500)), (int(rng.integers(1, 8)), int(rng.integers(1, 40)))
    ta, ot = TimeAxis.daily("2001-01-01", T, "noleap"), OTime.noleap(2001, T)
    x = rng.normal(280, 8, (T,) + shape)
    x[rng.random(x.shape) < rng.choice([0.0, 0.02, 0.4])] = np.nan
    freq = str(rng.choice(["YS", "MS", "QS-DEC"]))
    if it % 3 == 1:
        op = str(rng.choice([">", ">=", "<", "<="]))
        thr = float(x[np.isfinite(x)][0]) if rng.random() < 0.5 and np.isfinite(x).any() else 281.0
        if not np.array_equal(hgen.threshold_count(x, op, thr, ta, freq, device=dev), ogen.threshold_count(x, op, thr, ot, freq)):
            print(json.dumps({"FAIL": "count_f64", "T": T, "shape": shape, "op": op, "freq": freq}))
            sys.exit(1)
        stats["count"] += 1
    else:
        op = str(rng.choice(["sum", "mean", "min", "max", "std", "var", "count", "argmax", "argmin"]))
        got = hgen.select_resample_op(x, op, ta, freq, device=dev)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want = ogen.select_resample_op(x, op, ot, freq)
        ok = np.array_equal(got, want, equal_nan=True) if op in ("count", "argmax", "argmin", "min", "max") else np.allclose(got, want, rtol=1e-12, atol=0, equal_nan=True)
        if not ok:
            print(json.dumps({"FAIL": "reduce_f64", "T": T, "shape": shape, "op": op, "freq": freq}))
            sys.exit(1)
        stats["reduce"] += 1
print(json.dumps({"ok": True, "iterations": stats}))
