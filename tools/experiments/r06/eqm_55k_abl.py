"""Round 6: EQM training on 55 152 steps x 1440 x 90 under the XH_HIST_ABL bits (results wrong for most): where do the ~20 ms
go that the call spends beyond its five streaming passes per array?"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import bench  # noqa: E402
from xclim_amd import kernels as K  # noqa: E402
from xclim_amd._capi import get_device  # noqa: E402

dev = get_device()
T, C = 55152, 1440 * 90
base = bench.seasonal_base(T)
ref = K.fill_synthetic(dev, T, C, 0, 4, base, 3.0)
hist = K.fill_synthetic(dev, T, C, 0, 5, base + np.float32(1.5), 3.3)
q = (np.arange(20) + 0.5) / 20
af, hq = dev.empty((20, C), np.float32), dev.empty((20, C), np.float32)
os.environ["XH_DIAGNOSTICS"] = "1"
os.environ["XH_HIST_STATS"] = "1"
out = {}
for abl in sys.argv[1:] or ["0"]:
    os.environ["XH_HIST_ABL"] = abl
    out.setdefault(abl, []).append(round(bench.event_time(dev, lambda: K.eqm_train(dev, ref, hist, q, "+", out=(af, hq)), 2), 3))
print(json.dumps(out))
