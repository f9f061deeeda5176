"""EQM train on 30 years (BASELINE configs[3]) alone, HIP-event time; used to tune the transposed-batch pipeline."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from xclim_amd import kernels as K
from xclim_amd._capi import Device

T = int(os.environ.get("XH_BENCH_T", "10950"))
C = int(sys.argv[1]) if len(sys.argv) > 1 else 1440 * 720
dev = Device(0)
base = bench.seasonal_base(T)
ref = K.fill_synthetic(dev, T, C, 0, 4, base, 3.0)
hist = K.fill_synthetic(dev, T, C, 0, 5, base + np.float32(1.5), 3.3)
q = (np.arange(20) + 0.5) / 20
af, hq = dev.empty((20, C), np.float32), dev.empty((20, C), np.float32)
ms = bench.event_time(dev, lambda: K.eqm_train(dev, ref, hist, q, "+", out=(af, hq)), 2)
print(json.dumps({"T": T, "C": C, "train_ms": ms, "GB/s": 8.0 * T * C / ms / 1e6, "prio": os.environ.get("XH_STREAM2_PRIO")}))
