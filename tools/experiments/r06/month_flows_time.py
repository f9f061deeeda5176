"""Round 6: the three mappings with group="time.month" (12 groups of ~930 rows) and group="time" — train / adjust wall clock,
30 years x 1440 x 90 (a check that no per-group loop is left launch- or host-bound)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
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
sim = K.fill_synthetic(dev, T, C, 0, 6, base + np.float32(3.5), 3.3)
out = {}
only = os.environ.get("ONLY", "")
for group in ("time.month", "time"):
    for cls in (sdba.EmpiricalQuantileMapping, sdba.QuantileDeltaMapping, sdba.DetrendedQuantileMapping):
        if only and f"{cls.__name__[:3]}_{group}" != only:
            continue
        tag = f"{cls.__name__[:3]}_{group}"
        for rep in range(2):
            dev.sync(); t0 = time.perf_counter()
            m = cls.train(ref, hist, nquantiles=20, kind="+", group=group, time=ta, device=dev)
            dev.sync(); out.setdefault(tag + "_train_ms", []).append(round((time.perf_counter() - t0) * 1e3, 1))
        for interp in ("nearest", "linear"):
            for rep in range(2):
                dev.sync(); t0 = time.perf_counter()
                s = m.adjust(sim, interp=interp, time=ta, keep=True) if group != "time" else m.adjust(sim, interp=interp, keep=True)
                dev.sync(); out.setdefault(f"{tag}_adjust_{interp}_ms", []).append(round((time.perf_counter() - t0) * 1e3, 1))
                del s
print(json.dumps(out))
