"""Fuzzing of percentile_doy on multi-year base periods (the top-16 register kernel with shared pair merges and its
wave-uniform fast path, the merge / LDS kernels for everything else) against the oracle's stacked calc_perc.
usage: python tools/fuzz_pdoy.py [seconds]"""
import json
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import calendar as ocal  # noqa: E402  (checker only)
from oracle import quantile as oq  # noqa: E402
from oracle.timeutil import OTime  # noqa: E402
from xclim_amd import kernels as K  # noqa: E402
from fuzzdev import get_fuzz_device  # noqa: E402
from xclim_amd.timeaxis import TimeAxis  # noqa: E402

dev = get_fuzz_device()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "4242")))
t_end = time.time() + budget
n = 0
while time.time() < t_end:
    nyears = int(rng.integers(7, 45))
    window = int(rng.choice([3, 5, 7]))
    calendar = str(rng.choice(["noleap", "standard", "360_day"]))
    C = int(rng.integers(1, 150))
    if calendar == "standard":
        T = 365 * nyears + (nyears + 3) // 4
        ta, ot = TimeAxis.daily("2000-01-01", T), OTime.standard("2000-01-01", T)
    elif calendar == "noleap":
        T = 365 * nyears
        ta, ot = TimeAxis.daily("2000-01-01", T, "noleap"), OTime.noleap(2000, T)
    else:
        T = 360 * nyears
        ta, ot = TimeAxis.daily("2000-01-01", T, "360_day"), OTime.noleap(2000, T, "360_day")
    t = np.arange(T)[:, None]
    x = (288 + 12 * np.sin(2 * np.pi * (t - 100) / 365) + rng.normal(0, 3, (T, C))).astype(np.float32)
    if rng.random() < 0.3:
        x = np.round(x, 1)  # ties
    x[rng.random((T, C)) < rng.choice([0.0, 0.0, 0.002, 0.05, 0.4])] = np.nan
    if C > 2:
        x[:, 0] = np.nan
        x[:, 1] = 280.0
    nper = int(rng.integers(1, 4))
    per = sorted(float(p) for p in rng.choice([0.0, 1.0, 5.0, 10.0, 25.0, 50.0, 75.0, 90.0, 95.0, 99.0, 100.0, 93.3], nper, replace=False))
    tb, years, doys = ta.doy_table()
    out = K.percentile_doy(dev, dev.to_device(x), tb, window, per).get()
    rr = ocal.rolling_construct_center(x, window)
    stack = np.full((len(doys), len(years), C, window), np.nan, dtype=np.float32)
    stack[np.searchsorted(doys, ot.doy), np.searchsorted(years, ot.year)] = rr
    stack = np.moveaxis(stack, 1, -2).reshape(len(doys), C, len(years) * window)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exp = np.moveaxis(oq.calc_perc(stack, per, 1 / 3, 1 / 3), -1, 0)
    if not np.allclose(out, exp, rtol=1e-12, atol=0, equal_nan=True):
        bad = np.argwhere(~(np.isclose(out, exp, rtol=1e-12, atol=0) | (np.isnan(out) & np.isnan(exp))))
        print(json.dumps({"FAIL": "pdoy", "nyears": nyears, "window": window, "calendar": calendar, "C": C, "per": per, "first": bad[:4].tolist(),
                          "got": out[tuple(bad[0])].item(), "exp": exp[tuple(bad[0])].item(), "n": n}))
        sys.exit(1)
    n += 1
print(json.dumps({"ok": True, "iterations": n}))
