"""Differential fuzzing of the round-5 kernels, BITWISE against the code they replace (same library, diagnostic switches):
  fused   k_hs_fused (both passes of a tile in one kernel, second pass in reverse row order; select4.hip) vs the two
          kernels of rounds 3-4 (XH_HIST_FUSED=0) vs the transposed selection pipeline (XH_SELECT_NOHIST), incl. grids of
          more tiles than workgroups (every workgroup walks several tiles: LDS state between tiles) and series beyond
          32768 steps (collect rounds > 0 behind the fused round 0)
  qdm     xh_qdm_hist (QDM "nearest" on long series: class boundaries as order statistics of the two-pass histogram
          selection + a streaming classification; select4.hip) vs the exact-rank kernels it replaces (XH_QDM_NOHIST: a
          column per workgroup behind transposes, or the global sort beyond 32768 steps): quantised fields (ties at the
          cut positions), dry days (a pure bin holds the minimum), NaN samples, NaN factors, both kinds / extrapolations
usage: python tools/fuzz_r05.py [seconds]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xclim_amd import kernels as K  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fuzzdev import get_fuzz_device  # noqa: E402

dev = get_fuzz_device()
# FUZZ_DEVICE=hostsim (tests/hostsim: the same kernels on CPU fibers): small grids, nothing beyond 32768 steps (the exact-rank side
# of that comparison is rocPRIM's global sort, not simulated), no grid of more tiles than workgroups
SMALL = os.environ.get("FUZZ_DEVICE") == "hostsim"
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "2025")))
os.environ["XH_DIAGNOSTICS"] = "1"
stats = {"fused": 0, "fused_long": 0, "fused_wide": 0, "qdm": 0, "qdm_long": 0}
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


def qdm_case(it):
    long_ = it % 6 == 5 and not SMALL
    T = int(rng.integers(32769, 60000)) if long_ else int(rng.integers(1025, 5000 if SMALL else 12000))
    C = int(rng.integers(1, 150 if long_ else (70 if SMALL else 500)))
    nq = int(rng.integers(2, 33))
    x = field(T, C, int(rng.integers(0, 4)))
    if rng.random() < 0.3:   # a coarse grid of values: ties everywhere, also at the extremes
        x = np.round(x, int(rng.integers(-1, 2))).astype(np.float32)
    q = np.unique(np.sort(rng.random(nq)) * 0.98 + 0.01)
    if len(q) < 2:
        return
    af = rng.normal(1.0, 0.4, (len(q), C)).astype(np.float32)
    af[rng.random(af.shape) < 0.1] = np.nan
    if C > 5:
        af[:, 4] = np.nan
        af[1:, 5] = np.nan
    kind, extrap = ("+", "*")[int(rng.integers(0, 2))], ("constant", "nan")[int(rng.integers(0, 2))]
    d_x, d_af = dev.to_device(x), dev.to_device(af)
    a = K.qdm_adjust(dev, d_x, d_af, q, kind, "nearest", extrap).get()
    b = with_env("XH_QDM_NOHIST", "1", lambda: K.qdm_adjust(dev, d_x, d_af, q, kind, "nearest", extrap).get())
    if not same(a, b):
        bad = np.argwhere(~((a == b) | (np.isnan(a) & np.isnan(b))))
        print(json.dumps({"FAIL": "qdm", "T": T, "C": C, "nq": len(q), "kind": kind, "extrap": extrap, "nbad": len(bad),
                          "first": bad[:4].tolist(), "cols": np.unique(bad[:, 1])[:8].tolist(), "it": it,
                          "a": a[tuple(bad[0])].item(), "b": b[tuple(bad[0])].item(), "x": x[tuple(bad[0])].item()}))
        sys.exit(1)
    stats["qdm_long" if long_ else "qdm"] += 1


only = os.environ.get("FUZZ_ONLY", "")
it = 0
while time.time() < t_end:
    it += 1
    if only == "qdm" or (only == "" and it % 2 == 0):
        qdm_case(it)
        continue
    which = (it // 2) % 8
    if SMALL and which < 2:
        which = 2
    if which == 0:     # more tiles than workgroups
        T, C, key = int(rng.integers(1025, 1400)), int(rng.integers(256 * 64 + 1, 3 * 256 * 64)), "fused_wide"
    elif which == 1:   # collect rounds behind the fused round 0
        T, C, key = int(rng.integers(32769, 60000)), int(rng.integers(1, 200)), "fused_long"
    else:
        T, C, key = int(rng.integers(1025, 4000 if SMALL else 9000)), int(rng.integers(1, 80 if SMALL else 600)), "fused"
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
