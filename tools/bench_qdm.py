"""QDM adjust (nearest, constant) on 365 x 1440 x 720: the cut-value kernel of qdm2.hip against the exact-rank pipeline."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from xclim_amd import kernels as K  # noqa: E402
from xclim_amd._capi import get_device  # noqa: E402

dev = get_device()
T, C = 365, 1440 * 720
base = bench.seasonal_base(T)
out = {}
for name, kind, fillkind in (("temperature", "+", 0), ("precipitation", "*", 1)):
    if fillkind == 0:
        sim = K.fill_synthetic(dev, T, C, 0, 6, base + np.float32(3.5), 3.3)
    else:
        sim = K.fill_synthetic(dev, T, C, 1, 7, np.zeros(T, np.float32), 40.0 / 86400.0, 0.3)
    q = (np.arange(20) + 0.5) / 20
    af = dev.to_device(np.random.default_rng(1).normal(1.0, 0.2, (20, C)).astype(np.float32))
    scen = dev.empty((T, C), np.float32)
    ms = bench.event_time(dev, lambda: K.qdm_adjust(dev, sim, af, q, kind, "nearest", "constant", out=scen), 5)
    out[name] = {"ms": ms, "GB/s": 8.0 * T * C / ms / 1e6, "frac": 8.0 * T * C / ms / 1e6 / 8000.0}
    os.environ["XH_DIAGNOSTICS"] = "1"
    os.environ["XH_QDM_NOREGSORT"] = "1"
    ref = dev.empty((T, C), np.float32)
    out[name]["legacy_ms"] = bench.event_time(dev, lambda: K.qdm_adjust(dev, sim, af, q, kind, "nearest", "constant", out=ref), 2)
    del os.environ["XH_QDM_NOREGSORT"]
    a, b = scen.get(), ref.get()
    out[name]["identical"] = bool(np.array_equal(a, b, equal_nan=True))
    for x in (sim, af, scen, ref):
        x.free()
print(json.dumps(out))
