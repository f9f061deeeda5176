"""tx90p on 30 years (10950 x C): two-step chain vs the fused count kernel, HIP-event times."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from xclim_amd import kernels as K
from xclim_amd._capi import Device
from xclim_amd.timeaxis import TimeAxis
T = 10950
C = int(sys.argv[1]) if len(sys.argv) > 1 else 1440 * 720
dev = Device(0)
ta = TimeAxis.daily("1981-01-01", T, "noleap")
tb, years, doys = ta.doy_table()
out = {"C": C}
tas = K.fill_synthetic(dev, T, C, 0, 2, bench.seasonal_base(T), 3.0)
per = dev.empty((1, len(doys), C), np.float64)
out["percentile_doy_ms"] = bench.event_time(dev, lambda: K.percentile_doy(dev, tas, tb, 5, [90.0], out=per), 2)
tidx = dev.to_device(np.searchsorted(doys, ta.doy).astype(np.int32))
for freq in ("YS", "MS"):
    seg, _ = ta.segments(freq)
    P = len(seg) - 1
    cnt, val = dev.empty((P, C), np.int32), dev.empty((P, C), np.int32)
    out[f"threshold_count_{freq}_ms"] = bench.event_time(dev, lambda: K.threshold_count(dev, tas, ">", seg, doy_table=per.reshape(len(doys), C), tidx=tidx, out=(cnt, val)), 2)
    period = (np.searchsorted(seg, tb, side="right") - 1).astype(np.int32)
    period[tb < 0] = -1
    c2, v2 = dev.empty((P, C), np.int32), dev.empty((P, C), np.int32)
    out[f"fused_{freq}_ms"] = bench.event_time(dev, lambda: K.percentile_doy_count(dev, tas, tb, 5, 90.0, ">", period, P, out=(c2, v2)), 2)
    out[f"equal_{freq}"] = bool(np.array_equal(c2.get(), cnt.get()) and np.array_equal(v2.get(), val.get()))
print(json.dumps(out))
