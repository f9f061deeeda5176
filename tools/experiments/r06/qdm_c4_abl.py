"""Round 6: where do the 6.5 ms go that k_hs_collect spends in QDM mode beyond the quantile mode (17.5 against 11.0 ms at
30 years x 1440 x 720)?  XH_HIST_ABL bits switch phases off (results wrong); kernel times from HIP events around the call."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import bench  # noqa: E402
from xclim_amd import kernels as K  # noqa: E402
from xclim_amd._capi import get_device  # noqa: E402

dev = get_device()
T, C = 10950, 1440 * 720
base = bench.seasonal_base(T)
sim = K.fill_synthetic(dev, T, C, 0, 6, base + np.float32(3.5), 3.3)
q = (np.arange(20) + 0.5) / 20
af = dev.to_device(np.random.default_rng(1).normal(1.0, 0.2, (20, C)).astype(np.float32))
scen = dev.empty((T, C), np.float32)
out = {}
os.environ["XH_DIAGNOSTICS"] = "1"
for abl in sys.argv[1:] or ["0"]:
    os.environ["XH_HIST_ABL"] = abl
    out.setdefault(abl, []).append(round(bench.event_time(dev, lambda: K.qdm_adjust(dev, sim, af, q, "+", "nearest", "constant", out=scen), 2), 3))
print(json.dumps(out))
