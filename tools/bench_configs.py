"""BASELINE.json configs 3 and 4 (and the 30-year tx90p half of config 5) at FULL size on one MI355X, HIP-event times.

  config 3: maximum_consecutive_dry_days on 3650 x 1440 x 720 pr (15.1 GB)
  config 4: EmpiricalQuantileMapping train + adjust, 30-yr daily 1440 x 720 ref/hist/sim (3 x 45.4 GB in, 45.4 GB out)
  tx90p-30: percentile_doy (150 samples per doy) + threshold_count on 10950 x 1440 x 720 (45.4 GB)
Inputs are generated on the device (never cross PCIe).  Run on the GPU box: python tools/bench_configs.py [--small]
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from xclim_amd import kernels as K  # noqa: E402
from xclim_amd._capi import Device  # noqa: E402
from xclim_amd.timeaxis import TimeAxis  # noqa: E402

small = "--small" in sys.argv
Y, X = (1440, 72) if small else (1440, 720)
C = Y * X
dev = Device(0)
PEAK = bench.HBM_PEAK_GBS
res = {"grid": [Y, X], "device": dev.name()}

# ---- config 3 ----
T = 3650
ta = TimeAxis.daily("2001-01-01", T, "noleap")
seg, _ = ta.segments("YS")
P = len(seg) - 1
pr = K.fill_synthetic(dev, T, C, 1, 3, np.zeros(T, np.float32), 40.0 / 86400.0, 0.3)
o, v = dev.empty((P, C), np.float32), dev.empty((P, C), np.int32)
ms = bench.event_time(dev, lambda: K.run_stats(dev, pr, "max", 1, seg, cut=True, fused_op="<", thresh=1.0 / 86400.0, out=(o, v)), 5)
E = float(T) * C
b = 4 * E + 8 * P * C
res["config3_cdd_3650"] = {"ms": ms, "GB/s": b / ms / 1e6, "frac": b / ms / 1e6 / PEAK, "cell-timesteps/s": E / ms * 1e3}
print(json.dumps(res["config3_cdd_3650"]), flush=True)
pr.free()

# ---- tx90p on 30 years ----
T = 10950
ta = TimeAxis.daily("1981-01-01", T, "noleap")
tb, years, doys = ta.doy_table()
seg, _ = ta.segments("YS")
P = len(seg) - 1
base = bench.seasonal_base(T)
tas = K.fill_synthetic(dev, T, C, 0, 2, base, 3.0)
per = dev.empty((1, len(doys), C), np.float64)
cnt, val = dev.empty((P, C), np.int32), dev.empty((P, C), np.int32)
tidx = dev.to_device(np.searchsorted(doys, ta.doy).astype(np.int32))
ms_p = bench.event_time(dev, lambda: K.percentile_doy(dev, tas, tb, 5, [90.0], out=per), 2)
ms_c = bench.event_time(dev, lambda: K.threshold_count(dev, tas, ">", seg, doy_table=per.reshape(len(doys), C), tidx=tidx, out=(cnt, val)), 3)
E = float(T) * C
bp, bc = 4 * E + 8 * len(doys) * C, 4 * E + 8 * len(doys) * C + 8 * P * C
res["tx90p_30yr"] = {"percentile_doy_ms": ms_p, "percentile_doy_GB/s": bp / ms_p / 1e6, "threshold_count_ms": ms_c,
                     "threshold_count_GB/s": bc / ms_c / 1e6, "total_ms": ms_p + ms_c, "GB/s": (bp + bc) / (ms_p + ms_c) / 1e6,
                     "frac": (bp + bc) / (ms_p + ms_c) / 1e6 / PEAK, "cell-timesteps/s": E / (ms_p + ms_c) * 1e3,
                     "exceedance_mean": float(cnt.get().mean())}
print(json.dumps(res["tx90p_30yr"]), flush=True)
per.free(); cnt.free(); val.free()

# ---- config 4: EQM train + adjust on 30 years ----
ref = tas
hist = K.fill_synthetic(dev, T, C, 0, 5, base + np.float32(1.5), 3.3)
q = (np.arange(20) + 0.5) / 20
af, hq = dev.empty((20, C), np.float32), dev.empty((20, C), np.float32)
ms_tr = bench.event_time(dev, lambda: K.eqm_train(dev, ref, hist, q, "+", out=(af, hq)), 1)
ref.free()
sim = K.fill_synthetic(dev, T, C, 0, 6, base + np.float32(3.5), 3.3)
scen = dev.empty((T, C), np.float32)
ms_ad = bench.event_time(dev, lambda: K.eqm_adjust(dev, sim, af, hq, "+", "nearest", "constant", out=scen), 2)
res["config4_eqm_30yr"] = {"train_ms": ms_tr, "train_GB/s": 8 * E / ms_tr / 1e6, "adjust_ms": ms_ad, "adjust_GB/s": 8 * E / ms_ad / 1e6,
                           "total_ms": ms_tr + ms_ad, "GB/s": 16 * E / (ms_tr + ms_ad) / 1e6, "frac": 16 * E / (ms_tr + ms_ad) / 1e6 / PEAK,
                           "cell-timesteps/s": E / (ms_tr + ms_ad) * 1e3}
print(json.dumps(res["config4_eqm_30yr"]), flush=True)
free, total = dev.mem_info()
res["hbm_used_GB_at_end"] = (total - free) / 1e9
print(json.dumps(res))
