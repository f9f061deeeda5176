"""Differential fuzzing of winsel.hip (round 6): EQM training with Grouper("time.dayofyear", window) through the sliding sorted
window (xh_eqm_train_window) BITWISE against the per-group selection it replaces (XH_WINSEL=0): random numbers of years, windows,
quantile counts, calendars (noleap / 360_day / standard with leap days), quantised fields (ties: equal samples leave and enter
together), NaN samples and whole NaN stretches, infinities, constant cells, series that start and end mid-year.
Four cases in ten train a DETRENDED mapping instead (xh_dqm_train_window: the window's mean, the picked samples normalised) against
its per-group chain (xh_poly_trend, xh_trend_apply, xh_eqm_train): the means agree to their summation order, so the tables are
compared for the same NaN / infinity pattern and to a few float32 ulps (cells with a negative or zero mean, kind "*": the order
reverses, samples drop out).
usage: python tools/fuzz_winsel.py [seconds]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xclim_amd import sdba  # noqa: E402
from xclim_amd.timeaxis import TimeAxis  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fuzzdev import get_fuzz_device  # noqa: E402

dev = get_fuzz_device()
SMALL = os.environ.get("FUZZ_DEVICE") == "hostsim"
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "2026")))
os.environ["XH_DIAGNOSTICS"] = "1"
t_end, it, served, n_dqm = time.time() + budget, 0, 0, 0
while time.time() < t_end:
    cal = str(rng.choice(["noleap", "360_day", "standard"]))
    years = int(rng.integers(2, 5 if SMALL else 33))
    window = int(rng.choice([3, 5, 7, 15, 31]))
    if years * window > 1000:
        window = max(3, (1000 // years) | 1)
    ylen = 360 if cal == "360_day" else 365
    T = years * ylen + int(rng.integers(-100, 100))
    start = f"{int(rng.integers(1990, 2010))}-{int(rng.integers(1, 13)):02d}-01" if rng.random() < 0.3 else "2000-01-01"
    ta = TimeAxis.daily(start, T, cal)
    cells = int(rng.integers(1, 6 if SMALL else 70))
    nq = int(rng.integers(1, 33))
    t = np.arange(T)[:, None]
    ref = (285 + 9 * np.sin(2 * np.pi * t / ylen) + rng.normal(0, 3, (T, cells))).astype(np.float32)
    hist = (ref[::-1] * 1.02 + rng.normal(0, 2, (T, cells))).astype(np.float32)
    mode = int(rng.integers(0, 5))
    if mode == 1:
        ref, hist = np.round(ref), np.round(hist, 1)
    if mode == 2:   # precipitation-like: many exact zeros
        ref = np.where(rng.random(ref.shape) < 0.6, 0.0, rng.gamma(0.8, 5.0, ref.shape)).astype(np.float32)
        hist = np.where(rng.random(ref.shape) < 0.5, 0.0, rng.gamma(0.9, 4.0, ref.shape)).astype(np.float32)
    for a in (ref, hist):
        a[rng.random(a.shape) < rng.choice([0.0, 0.01, 0.2])] = np.nan
        if rng.random() < 0.3:
            s0 = int(rng.integers(0, T - 10))
            a[s0:s0 + int(rng.integers(1, 400)), int(rng.integers(0, cells))] = np.nan
        if rng.random() < 0.2:
            a[:, int(rng.integers(0, cells))] = np.nan
        if rng.random() < 0.2:
            a[rng.integers(0, T, 5), int(rng.integers(0, cells))] = rng.choice([np.inf, -np.inf])
        if rng.random() < 0.1:
            a[:, int(rng.integers(0, cells))] = 7.0
    kind = "+" if mode != 2 else "*"
    dqm = rng.random() < 0.4
    if dqm:
        if rng.random() < 0.3:
            ref[:, int(rng.integers(0, cells))] *= -1.0
        if rng.random() < 0.2:
            hist[:, int(rng.integers(0, cells))] = 0.0
        if rng.random() < 0.5:
            kind = str(rng.choice(["+", "*"]))
    Model = sdba.DetrendedQuantileMapping if dqm else sdba.EmpiricalQuantileMapping
    entry = "xh_dqm_train_window" if dqm else "xh_eqm_train_window"
    os.environ["XH_WINSEL"] = "1"
    tr = dev.start_trace()
    a = Model.train(ref, hist, nquantiles=nq, kind=kind, group="time.dayofyear", window=window, time=ta, device=dev)
    dev.stop_trace()
    hit = any(n == entry for n, _ in tr)
    served += hit
    n_dqm += dqm and hit
    os.environ["XH_WINSEL"] = "0"
    b = Model.train(ref, hist, nquantiles=nq, kind=kind, group="time.dayofyear", window=window, time=ta, device=dev)
    if dqm:
        def close(x, y, rtol, atol):
            with np.errstate(all="ignore"):
                fin = np.isfinite(x) & np.isfinite(y)
                return (np.array_equal(np.isnan(x), np.isnan(y)) and np.array_equal(x[~fin & ~np.isnan(x)], y[~fin & ~np.isnan(y)])
                        and np.array_equal(fin, np.isfinite(y)) and bool((np.abs(x[fin] - y[fin]) <= atol + rtol * np.abs(y[fin])).all()))

        with np.errstate(all="ignore"):
            scale = max(1.0, float(np.nanmax(np.where(np.isfinite(ref), np.abs(ref), 0))), float(np.nanmax(np.where(np.isfinite(hist), np.abs(hist), 0))))
        ha, hb, fa, fb = a.hist_q, b.hist_q, a.af, b.af
        ok = close(a.scaling, b.scaling, 1e-12, 1e-13 * scale) and close(ha, hb, 3e-7, 1e-11 * scale)
        if ok and kind == "+":
            ok = close(fa, fb, 1e-6, 1e-6 * max(1.0, float(np.max(np.abs(hb[np.isfinite(hb)]), initial=0.0))))
        elif ok:    # (a ratio of two quantiles: compared where the denominator is not a rounding residue)
            big = np.isfinite(hb) & (np.abs(hb) > 1e-4) & np.isfinite(fb)
            ok = close(fa[big], fb[big], 3e-6, 1e-30)
    else:
        ok = np.array_equal(a.hist_q, b.hist_q, equal_nan=True) and np.array_equal(a.af, b.af, equal_nan=True)
    if not ok:
        bad = np.argwhere(~((a.hist_q == b.hist_q) | (np.isnan(a.hist_q) & np.isnan(b.hist_q))))
        print(json.dumps({"ok": False, "it": it, "dqm": bool(dqm), "kind": kind, "cal": cal, "years": years, "window": window, "T": T, "start": start,
                          "cells": cells, "nq": nq, "mode": mode, "first_bad": bad[:5].tolist()}))
        sys.exit(1)
    it += 1
print(json.dumps({"ok": True, "iterations": it, "through_the_sliding_kernel": served, "of_them_dqm": int(n_dqm)}))
