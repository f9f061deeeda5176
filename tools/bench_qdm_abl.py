"""Ablation of k_qdm_regsort (XH_QDM_ABL bits: 1 stats, 2 ranks, 4 sort, 8 adjacency, 16 picks, 32 apply)."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from xclim_amd import kernels as K
from xclim_amd._capi import get_device
dev = get_device()
T, C = 365, 1440 * 720
sim = K.fill_synthetic(dev, T, C, 1, 7, np.zeros(T, np.float32), 40.0 / 86400.0, 0.3)
q = (np.arange(20) + 0.5) / 20
af = dev.to_device(np.random.default_rng(1).normal(1.0, 0.2, (20, C)).astype(np.float32))
scen = dev.empty((T, C), np.float32)
os.environ["XH_DIAGNOSTICS"] = "1"
for abl in (0, 1, 2, 4, 8, 16, 32, 63, 59, 31):
    os.environ["XH_QDM_ABL"] = str(abl)
    ms = bench.event_time(dev, lambda: K.qdm_adjust(dev, sim, af, q, "*", "nearest", "constant", out=scen), 3)
    print(json.dumps({"abl": abl, "ms": ms}))
