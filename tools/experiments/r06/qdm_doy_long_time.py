"""Round 6: QuantileDeltaMapping.adjust / DetrendedQuantileMapping.adjust with 365 day-of-year groups on a LONG simulation
(YEARS x 365 steps x 1440 x 90; the model trained on 30 years): groups of YEARS rows."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import bench
from xclim_amd import kernels as K, sdba
from xclim_amd._capi import get_device
from xclim_amd.timeaxis import TimeAxis
dev = get_device()
years = int(sys.argv[1]) if len(sys.argv) > 1 else 151
T, C = 10950, 1440 * 90
ta = TimeAxis.daily("1981-01-01", T, "noleap")
base = bench.seasonal_base(T)
ref = K.fill_synthetic(dev, T, C, 0, 4, base, 3.0)
hist = K.fill_synthetic(dev, T, C, 0, 5, base + np.float32(1.5), 3.3)
out = {"years": years}
models = {}
for cls in (sdba.QuantileDeltaMapping, sdba.DetrendedQuantileMapping):
    models[cls.__name__] = cls.train(ref, hist, nquantiles=20, kind="+", group="time.dayofyear", window=31, time=ta, device=dev)
ref.free(); hist.free()
Ts = 365 * years
tas = TimeAxis.daily("1950-01-01", Ts, "noleap")
sim = K.fill_synthetic(dev, Ts, C, 0, 6, bench.seasonal_base(Ts) + np.float32(3.5), 3.3)
for name, m in models.items():
    for interp in ("nearest", "linear"):
        for rep in range(2):
            dev.sync(); t0 = time.perf_counter()
            s = m.adjust(sim, interp=interp, time=tas, keep=True)
            dev.sync(); out.setdefault(f"{name}_adjust_{interp}_ms", []).append(round((time.perf_counter() - t0) * 1e3, 1))
            del s
print(json.dumps(out))
