"""Differential fuzzing of the round-5 kernels, BITWISE against the code they replace (same library, diagnostic switches):
  fused   k_hs_fused (both passes of a tile in one kernel, second pass in reverse row order; select4.hip) vs the two
          kernels of rounds 3-4 (XH_HIST_FUSED=0) vs the transposed selection pipeline (XH_SELECT_NOHIST), incl. grids of
          more tiles than workgroups (every workgroup walks several tiles: LDS state between tiles) and series beyond
          32768 steps (collect rounds > 0 behind the fused round 0)
usage: python tools/fuzz_r05.py [seconds]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xclim_amd import kernels as K  # noqa: E402
from xclim_amd._capi import get_device  # noqa: E402

dev = get_device()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "2025")))
os.environ["XH_DIAGNOSTICS"] = "1"
stats = {"fused": 0, "fused_long": 0, "fused_wide": 0}
t_end = time.time() + budget


def with_env(name, val, fn):
    os.environ[name] = val
    try:
        return fn()
    finally:
        del os.environ[name]


def field(T, C, kind):
    if kind == 0:
        x = rng.normal(10, 4, (T, C))
    elif kind == 1:
        x = np.where(rng.random((T, C)) < rng.uniform(0.2, 0.8), 0.0, rng.gamma(0.7, 4.0, (T, C)))
    elif kind == 2:
        x = np.round(rng.normal(10, 4, (T, C)), int(rng.integers(0, 3)))       # quantised: many ties
    else:
        x = rng.normal(0, 1, (T, C)) * 10.0 ** rng.integers(-3, 4, (1, C))     # mixed scales, straddling zero
    x = x.astype(np.float32)
    x[rng.random((T, C)) < rng.choice([0.0, 0.01, 0.2])] = np.nan
    if C > 3:
        x[:, 0] = np.nan
        x[:, 1] = 3.5
        x[: T // 2, 2] = np.nan
    return x


def same(a, b):
    return np.array_equal(a, b, equal_nan=True)


it = 0
while time.time() < t_end:
    it += 1
    which = it % 8
    if which == 0:     # more tiles than workgroups
        T, C, key = int(rng.integers(1025, 1400)), int(rng.integers(256 * 64 + 1, 3 * 256 * 64)), "fused_wide"
    elif which == 1:   # collect rounds behind the fused round 0
        T, C, key = int(rng.integers(32769, 60000)), int(rng.integers(1, 200)), "fused_long"
    else:
        T, C, key = int(rng.integers(1025, 9000)), int(rng.integers(1, 600)), "fused"
    nq = int(rng.integers(1, 33))
    x = field(T, C, int(rng.integers(0, 4)))
    q = np.sort(rng.random(nq))
    if rng.random() < 0.3:
        q[0], q[-1] = 0.0, 1.0
    d_x = dev.to_device(x)
    a = K.quantile_series(dev, d_x, q).get()                                           # fused, reverse
    b = with_env("XH_HIST_FUSED", "0", lambda: K.quantile_series(dev, d_x, q).get())  # two kernels
    c = with_env("XH_HIST_FUSED", "1", lambda: K.quantile_series(dev, d_x, q).get())  # fused, forward
    ok = same(a, b) and same(a, c)
    if ok and T <= 32768 and which != 0:
        ok = same(a, with_env("XH_SELECT_NOHIST", "1", lambda: K.quantile_series(dev, d_x, q).get()))
    if not ok:
        bad = np.argwhere(~((a == b) | (np.isnan(a) & np.isnan(b))))
        print(json.dumps({"FAIL": key, "T": T, "C": C, "nq": nq, "first": bad[:4].tolist(), "it": it}))
        sys.exit(1)
    stats[key] += 1
    del d_x
print(json.dumps({"ok": True, "iterations": stats}))
