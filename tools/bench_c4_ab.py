"""A/B of select4's launch modes at config 4 inside ONE process (run-to-run noise of the box is +-1.5 ms on 40 ms):
XH_HIST_FUSED = 0 (two kernels) | 1 (fused, second pass forward) | 2 (fused, second pass in reverse), alternating."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from xclim_amd import kernels as K
from xclim_amd._capi import Device

os.environ["XH_DIAGNOSTICS"] = "1"
T = int(os.environ.get("XH_BENCH_T", "10950"))
C = int(sys.argv[1]) if len(sys.argv) > 1 else 1440 * 720
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["0", "1", "2"]
var = sys.argv[4] if len(sys.argv) > 4 else "XH_HIST_FUSED"
dev = Device(0)
base = bench.seasonal_base(T)
ref = K.fill_synthetic(dev, T, C, 0, 4, base, 3.0)
hist = K.fill_synthetic(dev, T, C, 0, 5, base + np.float32(1.5), 3.3)
q = (np.arange(20) + 0.5) / 20
af, hq = dev.empty((20, C), np.float32), dev.empty((20, C), np.float32)
res = {m: [] for m in modes}
for r in range(rounds):
    for m in modes:
        os.environ[var] = m
        res[m].append(bench.event_time(dev, lambda: K.eqm_train(dev, ref, hist, q, "+", out=(af, hq)), 2))
print(json.dumps({"T": T, "C": C, "var": var, "train_ms": {m: {"min": min(v), "median": float(np.median(v)), "all": [round(x, 2) for x in v]} for m, v in res.items()}}))
