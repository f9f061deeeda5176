"""Round 6: the sweep of distribution_sweep.py for the ADJUST side and the day-of-year flows — eqm_adjust (nearest / linear) with
tables trained on the same kind of field, the day-of-year training with a window (sliding sorted window) and without (one
launch), QDM adjust by day of the year — against the kelvin field of the same shape."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from xclim_amd import kernels as K, sdba
from xclim_amd._capi import get_device
from xclim_amd.timeaxis import TimeAxis
dev = get_device()
rng = np.random.default_rng(11)
C, T = 16384, 10950
ta = TimeAxis.daily("1981-01-01", T, "noleap")
q = (np.arange(20) + 0.5) / 20


def field(kind):
    t = np.arange(T, dtype=np.float32)[:, None]
    x = (288 + 12 * np.sin(2 * np.pi * (t - 100) / 365) + rng.normal(0, 3, (T, C))).astype(np.float32)
    if kind == "celsius":
        x -= np.float32(273.15)
    elif kind == "precip":
        x = np.where(rng.random((T, C)) < 0.6, 0.0, rng.gamma(0.8, 5.0, (T, C))).astype(np.float32)
    elif kind == "mask30":
        x[:, rng.random(C) < 0.3] = np.nan
    elif kind == "nan10":
        x[rng.random((T, C)) < 0.1] = np.nan
    elif kind == "ties":
        x = np.round(x, 1)
    return x


def timed(fn, n=2):
    r = fn(); dev.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    dev.sync()
    return round((time.perf_counter() - t0) / n * 1e3, 3), r


out = {}
for kind in ("kelvin", "celsius", "precip", "mask30", "nan10", "ties"):
    k = "*" if kind == "precip" else "+"
    ref, hist, sim = (dev.to_device(field(kind)) for _ in range(3))
    res = {}
    ms, (af, hq) = timed(lambda: K.eqm_train(dev, ref, hist, q, k))
    res["eqm_adjust_nearest"], _ = timed(lambda: K.eqm_adjust(dev, sim, af, hq, k, "nearest", "constant"))
    res["eqm_adjust_linear"], _ = timed(lambda: K.eqm_adjust(dev, sim, af, hq, k, "linear", "constant"))
    res["train_doy_w31"], m = timed(lambda: sdba.QuantileDeltaMapping.train(ref, hist, nquantiles=20, kind=k, group="time.dayofyear", window=31, time=ta, device=dev), 1)
    res["eqm_doy_adjust_nearest"], _ = timed(lambda: sdba.EmpiricalQuantileMapping.adjust(m, sim, interp="nearest", time=ta, keep=True))
    res["eqm_doy_adjust_linear"], _ = timed(lambda: sdba.EmpiricalQuantileMapping.adjust(m, sim, interp="linear", time=ta, keep=True))
    res["qdm_doy_adjust"], _ = timed(lambda: m.adjust(sim, interp="nearest", time=ta, keep=True))
    res["train_doy_nowindow"], _ = timed(lambda: sdba.EmpiricalQuantileMapping.train(ref, hist, nquantiles=20, kind=k, group="time.dayofyear", time=ta, device=dev))
    res["dqm_train_doy_w31"], d = timed(lambda: sdba.DetrendedQuantileMapping.train(ref, hist, nquantiles=20, kind=k, group="time.dayofyear", window=31, time=ta, device=dev), 1)
    res["dqm_doy_adjust"], _ = timed(lambda: d.adjust(sim, interp="nearest", time=ta, keep=True))
    out[kind] = res
    del ref, hist, sim, af, hq, m, d
worst = {f"{kd}.{op}": [ms, out["kelvin"][op], round(ms / out["kelvin"][op], 2)] for kd, v in out.items() for op, ms in v.items()
         if ms / out["kelvin"][op] > 1.5}
print(json.dumps({"ms": out, "slower_than_1.5x_kelvin": worst}))
