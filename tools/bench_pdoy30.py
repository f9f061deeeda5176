"""percentile_doy on 30 years (10950 x C), HIP-event time of the table kernel alone (diag envs read per launch)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from xclim_amd import kernels as K
from xclim_amd._capi import Device
from xclim_amd.timeaxis import TimeAxis
T = 10950
C = 1440 * 720
dev = Device(0)
ta = TimeAxis.daily("1981-01-01", T, "noleap")
tb, years, doys = ta.doy_table()
tas = K.fill_synthetic(dev, T, C, 0, 2, bench.seasonal_base(T), 3.0)
per = dev.empty((1, len(doys), C), np.float64)
pers = [float(p) for p in os.environ.get("PERS", "90").split(",")]
print(json.dumps({"ms": bench.event_time(dev, lambda: K.percentile_doy(dev, tas, tb, 5, pers, out=per if len(pers) == 1 else None), 3),
                  "env": {k: v for k, v in os.environ.items() if k.startswith("XH_")}, "pers": pers}))
