"""Differential fuzzing of the round-3 kernels against the kernels they replace (same library, diagnostic switches):
qdm2.hip vs the exact-rank QDM kernel, tcount.hip vs the per-period threshold_count, select4.hip vs the transposed
selection pipeline.  Every comparison is BITWISE.  usage: python tools/fuzz_r03.py [seconds]"""
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
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "12345")))
os.environ["XH_DIAGNOSTICS"] = "1"
stats = {"qdm": 0, "tcount": 0, "select4": 0}
t_end = time.time() + budget


def with_env(name, fn):
    os.environ[name] = "1"
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


it = 0
while time.time() < t_end:
    it += 1
    which = it % 3
    if os.environ.get("FUZZ_ONLY") == "qdm":   # (FUZZ_DEVICE=hostsim: the tile count's legacy route and select4 are GPU-only)
        which = 0
    if which == 0:  # QDM nearest, one-year series
        T, C, nq = int(rng.integers(360, 367)), int(rng.integers(1, 700)), int(rng.integers(2, 37))
        sim = field(T, C, int(rng.integers(0, 4)))
        q = np.sort(rng.random(nq)) * 0.98 + 0.01
        q = np.unique(q)
        nq = len(q)
        if nq < 2:
            continue
        af = rng.normal(1.0, 0.3, (nq, C)).astype(np.float32)
        af[rng.random((nq, C)) < rng.choice([0.0, 0.05])] = np.nan
        kind, extrap = str(rng.choice(["+", "*"])), str(rng.choice(["constant", "nan"]))
        d_s, d_a = dev.to_device(sim), dev.to_device(af)
        a = K.qdm_adjust(dev, d_s, d_a, q, kind, "nearest", extrap).get()
        b = with_env("XH_QDM_NOREGSORT", lambda: K.qdm_adjust(dev, d_s, d_a, q, kind, "nearest", extrap).get())
        if not np.array_equal(a, b, equal_nan=True):
            bad = np.argwhere(~((a == b) | (np.isnan(a) & np.isnan(b))))
            cols = np.unique(bad[:, 1])
            info = [{"col": int(c_), "nbad": int((bad[:, 1] == c_).sum()), "n": int(np.isfinite(sim[:, c_]).sum()), "nmin": int((sim[:, c_] == np.nanmin(sim[:, c_])).sum()) if np.isfinite(sim[:, c_]).any() else 0,
                     "nvn": int(np.isfinite(af[:, c_]).sum()), "ndistinct": int(len(np.unique(sim[:, c_][np.isfinite(sim[:, c_])])))} for c_ in cols[:6]]
            print(json.dumps({"FAIL": "qdm", "T": T, "C": C, "nq": nq, "kind": kind, "extrap": extrap, "first": bad[:4].tolist(), "it": it, "ncols_bad": int(len(cols)), "cols": info}))
            sys.exit(1)
        stats["qdm"] += 1
    elif which == 1:  # threshold_count, per-doy table, multi-year
        ndoy = int(rng.choice([360, 365, 366]))
        T, C = int(rng.integers(3 * ndoy + 1, 12 * ndoy)), int(rng.integers(64, 900))
        x = field(T, C, int(rng.integers(0, 4)))
        table = (10 + rng.normal(0, 4, (ndoy, C))).astype(np.float64)
        sel = rng.integers(0, T, 50)
        table[sel % ndoy, rng.integers(0, C, 50)] = x[sel, rng.integers(0, C, 50)].astype(np.float64)
        table[rng.random((ndoy, C)) < 0.001] = np.nan
        tidx = (np.arange(T) + int(rng.integers(0, ndoy))) % ndoy
        tidx = tidx.astype(np.int32)
        nper = int(rng.integers(1, 150))
        cuts = np.unique(np.concatenate([[0, T], rng.integers(0, T + 1, nper)]))
        if rng.random() < 0.5:
            cuts = cuts[1:-1] if len(cuts) > 3 else cuts
        seg = cuts.astype(np.int64)
        op = str(rng.choice([">", "<", ">=", "<="]))
        d_x, d_t = dev.to_device(x), dev.to_device(table)
        c1, v1 = K.threshold_count(dev, d_x, op, seg, doy_table=d_t, tidx=tidx)
        c1, v1 = c1.get(), v1.get()
        c2, v2 = with_env("XH_TCOUNT_LEGACY", lambda: tuple(o.get() for o in K.threshold_count(dev, d_x, op, seg, doy_table=d_t, tidx=tidx)))
        if not (np.array_equal(c1, c2) and np.array_equal(v1, v2)):
            bc, bv = np.argwhere(c1 != c2), np.argwhere(v1 != v2)
            print(json.dumps({"FAIL": "tcount", "T": T, "C": C, "P": len(seg) - 1, "op": op, "it": it, "ndoy": ndoy, "seg": seg[:6].tolist(),
                              "count_diff": bc[:4].tolist(), "valid_diff": bv[:4].tolist(),
                              "c1": c1[tuple(bc[0])].item() if len(bc) else None, "c2": c2[tuple(bc[0])].item() if len(bc) else None,
                              "v1": v1[tuple(bv[0])].item() if len(bv) else None, "v2": v2[tuple(bv[0])].item() if len(bv) else None,
                              "ncount_diff": len(bc), "nvalid_diff": len(bv)}))
            sys.exit(1)
        stats["tcount"] += 1
    else:  # long-series quantiles
        T, C, nq = int(rng.integers(1025, 9000)), int(rng.integers(1, 400)), int(rng.integers(1, 33))
        x = field(T, C, int(rng.integers(0, 4)))
        q = np.sort(rng.random(nq))
        if rng.random() < 0.3:
            q[0], q[-1] = 0.0, 1.0
        d_x = dev.to_device(x)
        a = K.quantile_series(dev, d_x, q).get()
        b = with_env("XH_SELECT_NOHIST", lambda: K.quantile_series(dev, d_x, q).get())
        if not np.array_equal(a, b, equal_nan=True):
            bad = np.argwhere(~((a == b) | (np.isnan(a) & np.isnan(b))))
            print(json.dumps({"FAIL": "select4", "T": T, "C": C, "nq": nq, "first": bad[:4].tolist(), "it": it}))
            sys.exit(1)
        stats["select4"] += 1
print(json.dumps({"ok": True, "iterations": stats}))
