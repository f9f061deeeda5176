import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from xclim_amd import kernels as K
from xclim_amd._capi import get_device
dev = get_device()
T, C = 365, 1440 * 720 // 4
base = bench.seasonal_base(T)
for name, kind, fk in (("temperature", "+", 0), ("precipitation", "*", 1)):
    sim = K.fill_synthetic(dev, T, C, 0, 6, base + np.float32(3.5), 3.3) if fk == 0 else K.fill_synthetic(dev, T, C, 1, 7, np.zeros(T, np.float32), 40.0 / 86400.0, 0.3)
    q = (np.arange(20) + 0.5) / 20
    afh = np.random.default_rng(1).normal(1.0, 0.2, (20, C)).astype(np.float32)
    af = dev.to_device(afh)
    a = K.qdm_adjust(dev, sim, af, q, kind, "nearest", "constant").get()
    os.environ["XH_DIAGNOSTICS"] = "1"; os.environ["XH_QDM_NOREGSORT"] = "1"
    b = K.qdm_adjust(dev, sim, af, q, kind, "nearest", "constant").get()
    del os.environ["XH_QDM_NOREGSORT"]
    x = sim.get()
    bad = ~((a == b) | (np.isnan(a) & np.isnan(b)))
    cols = np.flatnonzero(bad.any(axis=0))
    print(name, "mismatching elements", int(bad.sum()), "columns", len(cols), "tile-lane stats", np.bincount(cols % 32, minlength=32).tolist())
    for c in cols[:6]:
        rows = np.flatnonzero(bad[:, c])
        col = x[:, c]
        srt = np.sort(col)
        dup = np.flatnonzero(np.diff(srt) == 0)
        print(" col", int(c), "rows", rows[:8].tolist(), "nbad", len(rows), "n", int(np.isfinite(col).sum()), "ndup", len(dup), "dup ranks", dup[:6].tolist(),
              "ranks of bad", np.searchsorted(srt, col[rows[:8]]).tolist(), "a", a[rows[:3], c].tolist(), "b", b[rows[:3], c].tolist(), "x", col[rows[:3]].tolist())
