"""k_select_lean alone (time-minor input, no transposes) vs the time-major pipeline, T = 10950."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from xclim_amd import kernels as K
from xclim_amd._capi import Device
T, C = 10950, int(sys.argv[1]) if len(sys.argv) > 1 else 121600
dev = Device(0)
x = K.fill_synthetic(dev, T, C, 0, 4, bench.seasonal_base(T), 3.0)
xT = K.transpose(dev, x)
q = (np.arange(20) + 0.5) / 20
ms_minor = bench.event_time(dev, lambda: K.quantile_series(dev, xT, q, time_axis=1), 3)
ms_major = bench.event_time(dev, lambda: K.quantile_series(dev, x, q), 3)
print(json.dumps({"T": T, "C": C, "NT": os.environ.get("XH_LEAN_NT"), "select_only_ms_per_12160_cols": ms_minor * 12160 / C,
                  "pipeline_ms_per_12160_cols": ms_major * 12160 / C}))
