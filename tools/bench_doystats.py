import sys, os, json, numpy as np
sys.path.insert(0, "/root/repo")
import bench
from xclim_amd import kernels as K
from xclim_amd._capi import Device
from xclim_amd.timeaxis import TimeAxis
dev = Device(0)
for cal in ("noleap", "standard"):
    T = 10950 if cal == "noleap" else 10957
    C = 103680
    ta = TimeAxis.daily("1981-01-01", T, cal)
    tb, years, doys = ta.doy_table()
    x = K.fill_synthetic(dev, T, C, 0, 2, bench.seasonal_base(T), 3.0)
    ms = bench.event_time(dev, lambda: K.doy_mean_std(dev, x, tb, 5), 3)
    print(json.dumps({"cal": cal, "nyears": int(tb.shape[0]), "C": C, "ms": ms, "GB/s": 4.0 * T * C / ms / 1e6}))
