"""Oracle fuzzing with INFINITIES in the data (both signs, isolated and whole columns) — the lerp of numpy's quantile
turns inf - inf into NaN and the reference then falls back to nanmax (utl:552-554); sdba interpolations propagate the
NaN.  usage: python tools/fuzz_inf.py [seconds]

  nan_quantile (fp64 result)                 one-shot selection, any sample count
  quantile_series (fp32 nodes)               select3 (365), lean / histogram (select2/4), radix (select5)
  percentile_doy                             one year (k_pdoy_slide / LDS), multi-year (quad / top16 / merge ring)
  eqm train + adjust, qdm adjust             nearest / linear / cubic, constant / nan extrapolation
"""
import json
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import calendar as ocal  # noqa: E402
from oracle import quantile as oq  # noqa: E402
from oracle import sdba as osdba  # noqa: E402
from oracle.timeutil import OTime  # noqa: E402
from xclim_amd import kernels as K  # noqa: E402
from fuzzdev import get_fuzz_device  # noqa: E402
from xclim_amd.timeaxis import TimeAxis  # noqa: E402

warnings.simplefilter("ignore")
np.seterr(all="ignore")
dev = get_fuzz_device()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "777")))
stats = {"nan_quantile": 0, "quantile_series": 0, "percentile_doy": 0, "eqm": 0, "qdm": 0}
t_end = time.time() + budget


def field(T, C):
    x = rng.normal(10, 4, (T, C)).astype(np.float32)
    if rng.random() < 0.5:
        x = np.round(x, 1)
    p = float(rng.choice([0.002, 0.02, 0.3]))
    r = rng.random((T, C))
    x[r < p] = np.inf
    x[(r >= p) & (r < 2 * p)] = -np.inf
    x[rng.random((T, C)) < float(rng.choice([0.0, 0.01, 0.1]))] = np.nan
    if C > 6:
        x[:, 0] = np.inf
        x[:, 1] = -np.inf
        x[::2, 2] = np.inf
        x[1::2, 2] = -np.inf
        x[:, 3] = np.nan
        x[: T // 2, 4] = np.inf
        x[5:, 5] = np.nan  # five samples left
    return x


def check(name, got, exp, what, rtol=0.0):
    ok = np.allclose(got, exp, rtol=rtol, atol=0, equal_nan=True) if rtol else np.array_equal(got, exp, equal_nan=True)
    if not ok:
        g, e = np.asarray(got, dtype=np.float64), np.asarray(exp, dtype=np.float64)
        bad = np.argwhere(~((g == e) | (np.isnan(g) & np.isnan(e)) | (np.isfinite(g) & np.isfinite(e) & (np.abs(g - e) <= rtol * np.abs(e)))))
        print(json.dumps({"FAIL": name, "what": what, "n_bad": int(len(bad)), "first": bad[:5].tolist(),
                          "got": float(g[tuple(bad[0])]), "exp": float(e[tuple(bad[0])])}))
        sys.exit(1)
    stats[name] += 1


it = 0
while time.time() < t_end:
    it += 1
    which = it % 5
    if which == 0:
        n, C = int(rng.choice([3, 17, 150, 365, 1200])), int(rng.integers(7, 120))
        x = field(n, C)
        q = np.sort(rng.random(int(rng.integers(1, 6))))
        ab = (1.0, 1.0) if rng.random() < 0.5 else (1 / 3, 1 / 3)
        exp = oq.nan_quantile(x, q, 0, *ab)
        check("nan_quantile", K.nan_quantile(dev, dev.to_device(x), q, *ab).get(), exp, f"n={n} C={C} ab={ab}", rtol=1e-12)
    elif which == 1:
        T = int(rng.choice([365, 366, 360, 800, 3650, int(rng.integers(1025, 9000)), 40000 if it % 25 == 1 else 2000]))
        C = int(rng.integers(7, 100))
        x = field(T, C)
        q = np.sort(rng.random(int(rng.integers(1, 21))))
        exp = oq.nan_quantile(x, q, axis=0, alpha=1.0, beta=1.0).astype(np.float32)
        check("quantile_series", K.quantile_series(dev, dev.to_device(x), q).get(), exp, f"T={T} C={C} nq={len(q)}")
    elif which == 2:
        nyears = int(rng.choice([1, 1, 3, 8, 30, 33]))
        cal = str(rng.choice(["standard", "noleap"]))
        T = 365 * nyears + ((nyears + 3) // 4 if cal == "standard" else 0)
        C = int(rng.integers(7, 90))
        x = field(T, C)
        if cal == "standard":
            ta, ot = TimeAxis.daily("2000-01-01", T), OTime.standard("2000-01-01", T)
        else:
            ta, ot = TimeAxis.daily("2000-01-01", T, "noleap"), OTime.noleap(2000, T, "noleap")
        tb, years, doys = ta.doy_table()
        w = int(rng.choice([3, 5, 5, 7]))
        per = sorted(float(v) for v in rng.choice([1.0, 5.0, 10.0, 50.0, 90.0, 95.0, 99.0], size=int(rng.integers(1, 3)), replace=False))
        try:
            got = K.percentile_doy(dev, dev.to_device(x), tb, w, per).get()
        except Exception as e:  # noqa: BLE001
            # (FUZZ_DEVICE=hostsim: an entry point the simulation lacks — none of this fuzzer's since the register percentile
            #  kernels run on fibers; kept as a guard)
            if os.environ.get("FUZZ_DEVICE") == "hostsim" and getattr(e, "code", None) == -5:
                stats["not_simulated"] = stats.get("not_simulated", 0) + 1
                continue
            raise
        rr = ocal.rolling_construct_center(x, w)
        stack = np.full((len(doys), len(years), C, w), np.nan, dtype=np.float32)
        stack[np.searchsorted(doys, ot.doy), np.searchsorted(years, ot.year)] = rr
        stack = np.moveaxis(stack, 1, -2).reshape(len(doys), C, len(years) * w)
        exp = np.moveaxis(oq.calc_perc(stack, per, 1 / 3, 1 / 3), -1, 0)
        check("percentile_doy", got, exp, f"nyears={nyears} cal={cal} w={w} per={per} C={C}", rtol=1e-12)
    elif which == 3:
        T, C, nq = int(rng.choice([365, 900, 3650])), int(rng.integers(7, 60)), int(rng.choice([5, 20, 50]))
        kind = str(rng.choice(["+", "*"]))
        q = osdba.equally_spaced_nodes(nq)
        # training on series WITH infinities: nodes and factors against the oracle
        ref, hist = field(T, C), field(T, C)
        af_e, hq_e = osdba.eqm_train(ref, hist, nq, kind)
        d_af, d_hq = K.eqm_train(dev, dev.to_device(ref), dev.to_device(hist), q, kind)
        check("eqm", d_hq.get(), hq_e.astype(np.float32), f"hist_q T={T} nq={nq}")
        # (a zero node: numpy's sort leaves the order of -0.0 and +0.0 to its algorithm, so the SIGN of x / 0 is not defined upstream)
        zero = hq_e == 0
        check("eqm", np.where(zero, 0, d_af.get()), np.where(zero, 0, af_e.astype(np.float32)), f"af T={T} nq={nq} kind={kind}", rtol=1e-6)
        # adjustment of a series with infinities through FINITE nodes (scipy's interp1d on infinite nodes is its own subject)
        ref, hist, sim = rng.normal(10, 4, (T, C)).astype(np.float32), rng.normal(11, 5, (T, C)).astype(np.float32), field(T, C)
        d_af, d_hq = K.eqm_train(dev, dev.to_device(ref), dev.to_device(hist), q, kind)
        af_h, hq_h = d_af.get(), d_hq.get()
        for interp in ("nearest", "linear", "cubic") if nq <= 32 else ("nearest", "linear"):
            ex = str(rng.choice(["constant", "nan"]))
            exp = osdba.eqm_adjust(sim, af_h, hq_h, kind, interp, ex).astype(np.float32)
            got = K.eqm_adjust(dev, dev.to_device(sim), d_af, d_hq, kind, interp, ex).get()
            check("eqm", got, exp, f"adjust {interp} {ex} {kind} T={T} nq={nq}", rtol=2e-6 if interp == "cubic" else 1e-6)
    else:
        T, C, nq = int(rng.choice([365, 1000, 3650])), int(rng.integers(7, 60)), int(rng.choice([5, 20, 50]))
        sim = field(T, C)
        q = osdba.equally_spaced_nodes(nq)
        af = rng.normal(1.0, 0.3, (nq, C)).astype(np.float32)
        af[rng.random(af.shape) < 0.05] = np.nan
        kind, interp, ex = str(rng.choice(["+", "*"])), str(rng.choice(["nearest", "linear"])), str(rng.choice(["constant", "nan"]))
        exp = osdba.qdm_adjust(sim, af, q, kind, interp, ex).astype(np.float32)
        got = K.qdm_adjust(dev, dev.to_device(sim), dev.to_device(af), q, kind, interp, ex).get()
        check("qdm", got, exp, f"{kind} {interp} {ex} T={T} nq={nq}", rtol=1e-6)

print(json.dumps({"ok": True, "iterations": it, "cases": stats}))
