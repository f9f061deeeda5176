"""Round 6: the three mappings with group="time.dayofyear" WITHOUT a window (365 groups of 30 rows), 30 years x 1440 x 90:
train wall clock through the one-launch kernels (XH_TRAIN_GROUPS=1) and through the per-group path it replaces (XH_TRAIN_GROUPS=0)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
os.environ["XH_DIAGNOSTICS"] = "1"
import bench
from xclim_amd import kernels as K, sdba
from xclim_amd._capi import get_device
from xclim_amd.timeaxis import TimeAxis
dev = get_device()
T, C = 10950, 1440 * 90
ta = TimeAxis.daily("1981-01-01", T, "noleap")
base = bench.seasonal_base(T)
ref = K.fill_synthetic(dev, T, C, 0, 4, base, 3.0)
hist = K.fill_synthetic(dev, T, C, 0, 5, base + np.float32(1.5), 3.3)
out = {}
for cls in (sdba.EmpiricalQuantileMapping, sdba.DetrendedQuantileMapping):
    for sw in ("1", "0"):
        os.environ["XH_TRAIN_GROUPS"] = sw
        for rep in range(2):
            dev.sync(); t0 = time.perf_counter()
            m = cls.train(ref, hist, nquantiles=20, kind="+", group="time.dayofyear", time=ta, device=dev)
            dev.sync(); out.setdefault(f"{cls.__name__}_train_groups{sw}_ms", []).append(round((time.perf_counter() - t0) * 1e3, 1))
print(json.dumps(out))
