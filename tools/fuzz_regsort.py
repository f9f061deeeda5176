"""One-year quantile_series / eqm_train: the register-sort kernel (select3.hip) against the histogram kernels of select.hip
(XH_SELECT_NOREGSORT=1), bitwise, and both layouts of the input.  usage: python tools/fuzz_regsort.py [seconds]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from xclim_amd import kernels as K  # noqa: E402
from fuzzdev import get_fuzz_device  # noqa: E402

dev = get_fuzz_device()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "5150")))
os.environ["XH_DIAGNOSTICS"] = "1"
t_end = time.time() + budget
n = 0
while time.time() < t_end:
    T, C, nq = int(rng.integers(360, 367)), int(rng.integers(1, 600)), int(rng.integers(1, 65))
    kind = int(rng.integers(0, 4))
    if kind == 0:
        x = rng.normal(10, 4, (T, C))
    elif kind == 1:
        x = np.where(rng.random((T, C)) < rng.uniform(0.2, 0.9), 0.0, rng.gamma(0.7, 4.0, (T, C)))
    elif kind == 2:
        x = np.round(rng.normal(10, 4, (T, C)), int(rng.integers(0, 3)))
    else:
        x = rng.normal(0, 1, (T, C)) * 10.0 ** rng.integers(-3, 4, (1, C))
    x = x.astype(np.float32)
    x[rng.random((T, C)) < rng.choice([0.0, 0.01, 0.3])] = np.nan
    if C > 3:
        x[:, 0] = np.nan
        x[:, 1] = -2.5
        x[1:, 2] = np.nan
    q = np.sort(rng.random(nq))
    if rng.random() < 0.3:
        q[0], q[-1] = 0.0, 1.0
    d = dev.to_device(x)
    a = K.quantile_series(dev, d, q).get()
    os.environ["XH_SELECT_NOREGSORT"] = "1"
    b = K.quantile_series(dev, d, q).get()
    del os.environ["XH_SELECT_NOREGSORT"]
    if not np.array_equal(a, b, equal_nan=True):
        bad = np.argwhere(~((a == b) | (np.isnan(a) & np.isnan(b))))
        print(json.dumps({"FAIL": "regsort", "T": T, "C": C, "nq": nq, "kind": kind, "first": bad[:4].tolist(), "a": a[tuple(bad[0])].item(), "b": b[tuple(bad[0])].item()}))
        sys.exit(1)
    n += 1
print(json.dumps({"ok": True, "iterations": n}))
