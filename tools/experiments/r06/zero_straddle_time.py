"""Round 6: do the selection kernels care where zero is?  The same synthetic field in kelvin (288 +- 12 + noise) and in degrees
Celsius (15 +- 12 + noise: winter values straddle zero) and as anomalies (0 +- 12): eqm_train wall clock for T = 365 (register
sort), 930 rows (k_select_quantile: a month group / a gathered day-of-year window) and 10950 (two-pass streaming histogram)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import bench
from xclim_amd import kernels as K
from xclim_amd._capi import get_device
dev = get_device()
C = 1440 * 90
q = (np.arange(20) + 0.5) / 20
out = {}
for T in (365, 930, 10950):
    for name, mean in (("kelvin", 288.0), ("celsius", 15.0), ("anomaly", 0.0)):
        base = bench.seasonal_base(T, mean=mean)
        ref = K.fill_synthetic(dev, T, C, 0, 4, base, 3.0)
        hist = K.fill_synthetic(dev, T, C, 0, 5, base + np.float32(1.5), 3.3)
        ts = []
        for rep in range(3):
            dev.sync(); t0 = time.perf_counter()
            r = K.eqm_train(dev, ref, hist, q, "+")
            dev.sync(); ts.append(round((time.perf_counter() - t0) * 1e3, 2))
        out[f"T{T}_{name}_ms"] = ts
        del ref, hist, r
# the exact-rank QDM kernel (k_qdm_columns: a histogram + a scan of the key's bin per key), interp="linear", 30 years on 1/8 band
T, Cq = 10950, C // 8
for name, mean in (("kelvin", 288.0), ("celsius", 15.0)):
    base = bench.seasonal_base(T, mean=mean)
    sim = K.fill_synthetic(dev, T, Cq, 0, 6, base, 3.3)
    af = dev.to_device(np.random.default_rng(1).normal(0, 1, (20, Cq)).astype(np.float32))
    ts = []
    for rep in range(2):
        dev.sync(); t0 = time.perf_counter()
        r = K.qdm_adjust(dev, sim, af, q, "+", "linear", "constant")
        dev.sync(); ts.append(round((time.perf_counter() - t0) * 1e3, 2))
    out[f"qdm_linear_T{T}_{name}_ms"] = ts
    del sim, af, r
print(json.dumps(out))
