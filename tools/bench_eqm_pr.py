"""quantile_series / eqm_train on precipitation-like series (70 % exact zeros -> one heavily tied histogram bin)
against temperature-like ones.  Run on the GPU box: python tools/bench_eqm_pr.py [T] [C]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xclim_amd import kernels as K
from xclim_amd._capi import Device
import bench

T = int(sys.argv[1]) if len(sys.argv) > 1 else 365
C = int(sys.argv[2]) if len(sys.argv) > 2 else 1440 * 720
dev = Device(0)
q = (np.arange(20) + 0.5) / 20
zero = np.zeros(T, dtype=np.float32)
for name, x in (("tas", K.fill_synthetic(dev, T, C, 0, 4, bench.seasonal_base(T), 3.0)),
                ("pr p_wet=0.3", K.fill_synthetic(dev, T, C, 1, 3, zero, 40.0 / 86400.0, 0.3)),
                ("pr p_wet=0.6", K.fill_synthetic(dev, T, C, 1, 3, zero, 40.0 / 86400.0, 0.6))):
    ms = bench.event_time(dev, lambda: K.quantile_series(dev, x, q), 3)
    print(json.dumps({"series": name, "T": T, "C": C, "quantile_series_ms": round(ms, 3), "GB/s": round(4.0 * T * C / ms / 1e6, 1)}))
    del x
