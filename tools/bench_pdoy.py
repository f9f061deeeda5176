"""percentile_doy micro benchmark for multi-year base periods (HIP-event times).  Run on the GPU box."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xclim_amd import kernels as K
from xclim_amd._capi import Device
from xclim_amd.timeaxis import TimeAxis
import bench

ny = int(sys.argv[1]) if len(sys.argv) > 1 else 30
C = int(sys.argv[2]) if len(sys.argv) > 2 else 1440 * 72
cal = sys.argv[3] if len(sys.argv) > 3 else "noleap"
T = 365 * ny if cal == "noleap" else 365 * ny + (ny + 3) // 4
dev = Device(0)
ta = TimeAxis.daily("2000-01-01", T, cal)
tb, years, doys = ta.doy_table()
x = K.fill_synthetic(dev, T, C, 0, 2, bench.seasonal_base(T), 3.0)
out = dev.empty((1, len(doys), C), np.float64)
ms = bench.event_time(dev, lambda: K.percentile_doy(dev, x, tb, 5, [90.0], out=out), 2)
E = float(T) * C
print(json.dumps({"nyears": ny, "C": C, "calendar": cal, "ms": ms, "GB/s_alg": (4 * E + 8 * len(doys) * C) / ms / 1e6,
                  "cell-timesteps/s": E / ms * 1e3}))
