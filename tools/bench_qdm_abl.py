"""k_qdm_regsort phase ablation (XH_QDM_ABL bits: 1 stats, 2 ranks, 4 sort, 16 picks, 32 apply; results wrong) on the
temperature-like 365 x 1440 x 720 field: HIP-event time of the whole xh_qdm_adjust per setting."""
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
sim = K.fill_synthetic(dev, T, C, 0, 6, bench.seasonal_base(T) + np.float32(3.5), 3.3)
q = (np.arange(20) + 0.5) / 20
af = dev.to_device(np.random.default_rng(1).normal(1.0, 0.2, (20, C)).astype(np.float32))
scen = dev.empty((T, C), np.float32)
os.environ["XH_DIAGNOSTICS"] = "1"
out = {}
for abl in [int(a) for a in (sys.argv[1:] or ["0", "1", "2", "3", "4", "16", "7", "23", "0"])]:
    os.environ["XH_QDM_ABL"] = str(abl)
    out[f"abl_{abl}"] = round(bench.event_time(dev, lambda: K.qdm_adjust(dev, sim, af, q, "+", "nearest", "constant", out=scen), 5), 4)
print(json.dumps(out))
