"""Differential fuzzing of the grouped entry points of round 6 — xh_qdm_adjust_groups, xh_poly_trend_groups, xh_trend_apply_groups,
xh_eqm_train_groups, xh_dqm_train_groups
(all groups of a sub-grouping in ONE launch, a group = a list of rows) — BITWISE against the per-group calls they replace
(xh_qdm_adjust / xh_poly_trend_u / xh_trend_apply_u / xh_eqm_train [+ xh_poly_trend, xh_trend_apply for the detrended training] on
each group's gathered rows): random group counts and sizes (empty groups,
rows in no group, up to 64 rows per group for the rank kernel), NaN samples, ties and the two zeros, constant cells, infinities,
NaN factors (dropped nodes), 1 to 32 nodes, all kinds / interpolations / extrapolations / operations, cell counts around the
vector width.  usage: python tools/fuzz_groups.py [seconds]"""
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
SMALL = os.environ.get("FUZZ_DEVICE") == "hostsim"
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "2026")))
t_end, stats = time.time() + budget, {"qdm": 0, "trend": 0, "train": 0}


def same(a, b):
    return np.array_equal(a, b, equal_nan=True)


def fail(what, **kw):
    print(json.dumps({"ok": False, "what": what, **kw}))
    sys.exit(1)


it = 0
while time.time() < t_end:
    it += 1
    G = int(rng.integers(1, 8 if SMALL else 60))
    most = int(rng.choice([1, 3, 9, 31, 32, 33, 64])) if not SMALL else int(rng.choice([1, 3, 9]))
    sizes = rng.integers(0, most + 1, G)
    sizes[int(rng.integers(0, G))] = most
    spare = int(rng.integers(0, 10))
    T = int(sizes.sum()) + spare
    C = int(rng.choice([1, 3, 4, 63, 64, 256, 260, 1000])) if not SMALL else int(rng.choice([1, 4, 5]))
    order = rng.permutation(T)
    rows = order[:T - spare]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    # (a group's rows in time order, as the host lists them)
    rows = np.concatenate([np.sort(rows[offs[g]:offs[g + 1]]) for g in range(G)]) if len(rows) else rows
    x = rng.normal(3, 2, (T, C)).astype(np.float32)
    mode = int(rng.integers(0, 4))
    if mode == 1:
        x = np.round(x)
    if mode == 2:
        x = np.where(rng.random(x.shape) < 0.6, 0.0, x).astype(np.float32)
        x[rng.random(x.shape) < 0.1] = -0.0
    x[rng.random(x.shape) < rng.choice([0.0, 0.05, 0.4])] = np.nan
    if rng.random() < 0.3:
        x[:, int(rng.integers(0, C))] = 2.5
    if rng.random() < 0.2:
        x[:, int(rng.integers(0, C))] = np.nan
    if rng.random() < 0.2:
        x[int(rng.integers(0, T)), int(rng.integers(0, C))] = rng.choice([np.inf, -np.inf])
    d_x = dev.to_device(x)
    listed = np.zeros(T, dtype=bool)
    listed[rows] = True
    if it % 3 == 2:
        y = (x[::-1] * rng.choice([1.0, -1.0, 1.1]) + rng.normal(0, 1, x.shape)).astype(np.float32)
        if rng.random() < 0.2:
            y[:, int(rng.integers(0, C))] = 0.0
        d_y = dev.to_device(y)
        nq = int(rng.integers(1, 40))
        q = np.sort(rng.random(nq)) if rng.random() < 0.3 else (np.arange(nq) + 0.5) / nq
        kind = str(rng.choice(["+", "*"]))
        res = K.eqm_train_groups(dev, d_x, d_y, rows, offs, q, kind)
        resn = K.eqm_train_groups(dev, d_x, d_y, rows, offs, q, kind, normalised=True)
        if res is None or resn is None:
            fail("train: refused", G=G, most=most)
        af, hq = (a.get() for a in res)
        naf, nhq, sc, muh = (a.get() for a in resn)
        inv = "-" if kind == "+" else "/"
        for g in range(G):
            r = rows[offs[g]:offs[g + 1]]
            if not len(r):
                if not (np.isnan(hq[g]).all() and np.isnan(nhq[g]).all() and np.isnan(muh[g]).all()):
                    fail("train: an empty group is not NaN", g=g)
                continue
            xg, yg = K.select_rows(dev, d_x, r), K.select_rows(dev, d_y, r)
            a_g, h_g = K.eqm_train(dev, xg, yg, q, kind)
            if not (same(hq[g], h_g.get()) and same(af[g], a_g.get())):
                fail("eqm_train_groups", it=it, G=G, g=g, n=len(r), C=C, nq=nq, kind=kind, mode=mode)
            mu_x, _ = K.poly_trend(dev, xg, 0)
            mu_y, _ = K.poly_trend(dev, yg, 0)
            a_g, h_g = K.eqm_train(dev, K.trend_apply(dev, xg, mu_x, None, inv), K.trend_apply(dev, yg, mu_y, None, inv), q, kind)
            with np.errstate(all="ignore"):
                esc = mu_x.get() - mu_y.get() if kind == "+" else mu_x.get() / mu_y.get()
            if not (same(muh[g], mu_y.get()) and same(sc[g], esc) and same(nhq[g], h_g.get()) and same(naf[g], a_g.get())):
                fail("dqm_train_groups", it=it, G=G, g=g, n=len(r), C=C, nq=nq, kind=kind, mode=mode)
        stats["train"] += 1
    elif it % 3 == 1:
        nq = int(rng.integers(1, 33))
        q = np.sort(rng.random(nq)) if rng.random() < 0.3 else (np.arange(nq) + 0.5) / nq
        if len(np.unique(q)) < nq:
            continue
        af = rng.normal(1, 0.5, (G, nq, C)).astype(np.float32)
        af[rng.random(af.shape) < rng.choice([0.0, 0.1, 0.7])] = np.nan
        kind, interp, extrap = str(rng.choice(["+", "*", "factor"])), str(rng.choice(["nearest", "linear"])), str(rng.choice(["constant", "nan"]))
        d_af = dev.to_device(af)
        out = dev.to_device(np.full((T, C), -3.0, np.float32))
        got = K.qdm_adjust_groups(dev, d_x, rows, offs, d_af, q, kind, interp, extrap, out=out)
        if got is None:
            fail("qdm: refused", G=G, most=most, nq=nq)
        got = got.get()
        if not (got[~listed] == -3.0).all():
            fail("qdm: wrote a row of no group", G=G, T=T, C=C)
        for g in range(G):
            r = rows[offs[g]:offs[g + 1]]
            if len(r):
                exp = K.qdm_adjust(dev, K.select_rows(dev, d_x, r), dev.to_device(af[g]), q, kind, interp, extrap).get()
                if not same(got[r], exp):
                    fail("qdm", it=it, G=G, g=g, n=len(r), C=C, nq=nq, kind=kind, interp=interp, extrap=extrap, mode=mode)
        stats["qdm"] += 1
    else:
        u = rng.normal(0, 100, T)
        d_u = dev.to_device(u, dtype=np.float64)
        degree = int(rng.integers(0, 2))
        p0, p1 = K.poly_trend_groups(dev, d_x, rows, offs, d_u, degree)
        P0, P1 = p0.get(), (p1.get() if p1 is not None else None)
        op = str(rng.choice(list("+-*/")))
        out = dev.to_device(np.full((T, C), -3.0, np.float32))
        got = K.trend_apply_groups(dev, d_x, rows, offs, p0, p1, op, u=d_u, out=out).get()
        if not (got[~listed] == -3.0).all():
            fail("trend: wrote a row of no group", G=G, T=T, C=C)
        for g in range(G):
            r = rows[offs[g]:offs[g + 1]]
            if not len(r):
                continue
            blk = K.select_rows(dev, d_x, r)
            ug = dev.to_device(np.ascontiguousarray(u[r]), dtype=np.float64)
            q0, q1 = K.poly_trend(dev, blk, degree, u=ug)
            if not same(P0[g], q0.get()) or (degree and not same(P1[g], q1.get())):
                fail("poly_trend_groups", it=it, G=G, g=g, n=len(r), C=C, degree=degree)
            if not same(got[r], K.trend_apply(dev, blk, q0, q1, op, u=ug).get()):
                fail("trend_apply_groups", it=it, G=G, g=g, n=len(r), C=C, degree=degree, op=op)
        stats["trend"] += 1
print(json.dumps({"ok": True, "iterations": stats}))
