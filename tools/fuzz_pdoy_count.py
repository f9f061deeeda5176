"""xh_percentile_doy_count on multi-year periods: the table-kernel + tile-count route against the fused top-16 COUNT kernel
(XH_PDOY_COUNT_FUSED=1) and against the two-step chain (percentile_doy + threshold_count), bitwise.
usage: python tools/fuzz_pdoy_count.py [seconds]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xclim_amd import kernels as K  # noqa: E402
from xclim_amd._capi import get_device  # noqa: E402
from xclim_amd.timeaxis import TimeAxis  # noqa: E402

dev = get_device()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "2024")))
os.environ["XH_DIAGNOSTICS"] = "1"
t_end = time.time() + budget
n = {"routes_equal": 0, "fused_unavailable": 0, "not_served": 0}
while time.time() < t_end:
    nyears, window = int(rng.integers(7, 40)), int(rng.choice([3, 5, 7]))
    cal = str(rng.choice(["noleap", "360_day"]))
    ndays = 365 if cal == "noleap" else 360
    T, C = ndays * nyears, int(rng.integers(64, 500))
    ta = TimeAxis.daily("2001-01-01", T, cal)
    t = np.arange(T)[:, None]
    x = (288 + 12 * np.sin(2 * np.pi * (t - 100) / ndays) + rng.normal(0, 3, (T, C))).astype(np.float32)
    if rng.random() < 0.3:
        x = np.round(x, 1)
    x[rng.random((T, C)) < rng.choice([0.0, 0.0, 0.003])] = np.nan
    per, op, freq = float(rng.choice([90.0, 95.0, 10.0, 5.0, 99.0])), str(rng.choice([">", ">=", "<", "<="])), str(rng.choice(["YS", "MS", "QS-DEC"]))
    tb, years, doys = ta.doy_table()
    seg, _ = ta.segments(freq)
    P = len(seg) - 1
    period = (np.searchsorted(seg, tb, side="right") - 1).astype(np.int32)
    period[tb < 0] = -1
    d = dev.to_device(x)
    a = K.percentile_doy_count(dev, d, tb, window, per, op, period, P)
    if a is None:
        n["not_served"] += 1
        continue
    a = (a[0].get(), a[1].get())
    os.environ["XH_PDOY_COUNT_FUSED"] = "1"
    b = K.percentile_doy_count(dev, d, tb, window, per, op, period, P)
    del os.environ["XH_PDOY_COUNT_FUSED"]
    table = K.percentile_doy(dev, d, tb, window, [per])
    tidx = np.searchsorted(doys, ta.doy).astype(np.int32)
    c = K.threshold_count(dev, d, op, seg, doy_table=table.reshape(len(doys), C), tidx=tidx)
    c = (c[0].get(), c[1].get())
    ok = np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1])
    if b is not None:
        b = (b[0].get(), b[1].get())
        ok = ok and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        n["routes_equal"] += 1
    else:
        n["fused_unavailable"] += 1
    if not ok:
        print(json.dumps({"FAIL": "pdoy_count", "nyears": nyears, "window": window, "cal": cal, "C": C, "per": per, "op": op, "freq": freq}))
        sys.exit(1)
print(json.dumps({"ok": True, "iterations": n}))
