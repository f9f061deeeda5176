"""Round 6: eqm_train on a field with 30 % all-NaN cells (a land / sea mask), T = 10950 / 930 / 365 — run under rocprofv3
--kernel-trace --stats to see which kernel the 3-10 x of distribution_sweep.py goes to."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from xclim_amd import kernels as K
from xclim_amd._capi import get_device
dev = get_device()
rng = np.random.default_rng(7)
C = 16384
q = (np.arange(20) + 0.5) / 20
T = int(sys.argv[1]) if len(sys.argv) > 1 else 10950
x = (288 + rng.normal(0, 3, (T, C))).astype(np.float32)
if os.environ.get("MASK", "1") == "1":
    x[:, rng.random(C) < 0.3] = np.nan
d = dev.to_device(x)
for _ in range(5):
    r = K.quantile_series(dev, d, q)
dev.sync()
