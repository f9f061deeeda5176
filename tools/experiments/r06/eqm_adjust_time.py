"""Round 6: xh_eqm_adjust at 365 x 1440 x 720 (nearest, linear) and 10950 x 1440 x 90 — HIP-event times; run once per library
build by tools/experiments/r06/gpu_r06_k.sh (alternating processes on one box)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import bench  # noqa: E402
from xclim_amd import kernels as K  # noqa: E402
from xclim_amd._capi import get_device  # noqa: E402

dev = get_device()
out = {}
for T, C in ((365, 1440 * 720), (10950, 1440 * 90)):
    base = bench.seasonal_base(T)
    sim = K.fill_synthetic(dev, T, C, 0, 6, base + np.float32(3.5), 3.3)
    rng = np.random.default_rng(1)
    hq = dev.to_device(np.sort(rng.normal(288, 9, (20, C)).astype(np.float32), axis=0))
    af = dev.to_device(rng.normal(1.0, 0.2, (20, C)).astype(np.float32))
    scen = dev.empty((T, C), np.float32)
    for interp in ("nearest", "linear"):
        out[f"{T}_{interp}"] = round(bench.event_time(dev, lambda: K.eqm_adjust(dev, sim, af, hq, "+", interp, "constant", out=scen), 8), 4)
    out[f"{T}_sum"] = float(scen.get()[::37].astype(np.float64).sum())
    for a in (sim, hq, af, scen):
        a.free()
print(json.dumps(out))
