"""EQM adjust timing per interpolation (HIP events).  Run on the GPU box: python tools/bench_eqm_adjust.py"""
import sys
sys.path.insert(0, ".")
import numpy as np
import bench
from xclim_amd import kernels as K
from xclim_amd._capi import Device

dev = Device(0)
T, C = 365, 1440 * 720
base = bench.seasonal_base(T)
ref = K.fill_synthetic(dev, T, C, 0, 4, base, 3.0)
hist = K.fill_synthetic(dev, T, C, 0, 5, base + np.float32(1.5), 3.3)
q = (np.arange(20) + 0.5) / 20
af, hq = K.eqm_train(dev, ref, hist, q, "+")
for it in ("nearest", "linear", "cubic"):
    ms = bench.event_time(dev, lambda: K.eqm_adjust(dev, hist, af, hq, "+", it, "constant"), 5)
    print(it, round(ms, 4), "ms", round((8.0 * T * C + 160.0 * C) / ms / 1e6, 1), "GB/s", flush=True)
