"""EQM-only micro benchmark (HIP-event times) — used to tune the select kernels.  Run on the GPU box."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xclim_amd import kernels as K
from xclim_amd._capi import Device
import bench

T = int(sys.argv[1]) if len(sys.argv) > 1 else 365
C = int(sys.argv[2]) if len(sys.argv) > 2 else 1440 * 720
dev = Device(0)
base = bench.seasonal_base(T)
ref = K.fill_synthetic(dev, T, C, 0, 4, base, 3.0)
hist = K.fill_synthetic(dev, T, C, 0, 5, base + np.float32(1.5), 3.3)
q = (np.arange(20) + 0.5) / 20
af, hq = dev.empty((20, C), np.float32), dev.empty((20, C), np.float32)
ms_q = bench.event_time(dev, lambda: K.quantile_series(dev, ref, q), 3)
ms_tr = bench.event_time(dev, lambda: K.eqm_train(dev, ref, hist, q, "+", out=(af, hq)), 3)
refT = K.transpose(dev, ref)
ms_qT = bench.event_time(dev, lambda: K.quantile_series(dev, refT, q, time_axis=1), 3)
E = float(T) * C
print(json.dumps({"T": T, "C": C, "G": os.environ.get("XH_SELECT_G"), "quantile_tm_ms": ms_q, "GB/s": 4 * E / ms_q / 1e6,
                  "quantile_tminor_ms": ms_qT, "train_ms": ms_tr, "train GB/s": 8 * E / ms_tr / 1e6}))
