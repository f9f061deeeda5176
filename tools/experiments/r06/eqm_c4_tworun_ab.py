"""Round 6: EQM training at config 4, the candidate sort as TWO runs + merge-path picks (default) against one 512-slot network
(XH_HIST_ABL=2048: same results), alternating in one process; the outputs are compared bit for bit."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import bench  # noqa: E402
from xclim_amd import kernels as K  # noqa: E402
from xclim_amd._capi import get_device  # noqa: E402

dev = get_device()
T, C = 10950, 1440 * 720
base = bench.seasonal_base(T)
ref = K.fill_synthetic(dev, T, C, 0, 4, base, 3.0)
hist = K.fill_synthetic(dev, T, C, 0, 5, base + np.float32(1.5), 3.3)
q = (np.arange(20) + 0.5) / 20
os.environ["XH_DIAGNOSTICS"] = "1"
out = {"two_runs": [], "one_sort": []}
res = {}
for rnd in range(4):
    for name, abl in (("one_sort", "2048"), ("two_runs", "0")):
        os.environ["XH_HIST_ABL"] = abl
        af, hq = dev.empty((20, C), np.float32), dev.empty((20, C), np.float32)
        out[name].append(bench.event_time(dev, lambda: K.eqm_train(dev, ref, hist, q, "+", out=(af, hq)), 2))
        if rnd == 0:
            res[name] = (af.get(), hq.get())
        af.free(), hq.free()
out["identical"] = bool(np.array_equal(res["two_runs"][0], res["one_sort"][0], equal_nan=True) and np.array_equal(res["two_runs"][1], res["one_sort"][1], equal_nan=True))
print(json.dumps(out))
