"""Round 6: the headline kernels on fields that are NOT 288 +- 15 K — wall clock per call against the kelvin field of the same
shape (a pathology shows as a large ratio): degrees Celsius, precipitation-like (60 % exact zeros + gamma), 30 % of the cells
all-NaN (a land / sea mask), 10 % NaN samples, quantised to 0.1 (ties), a constant field."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from xclim_amd import kernels as K, calendar as xcal
from xclim_amd._capi import get_device
from xclim_amd.timeaxis import TimeAxis
dev = get_device()
rng = np.random.default_rng(7)
C = 16384
q = (np.arange(20) + 0.5) / 20


def field(kind, T):
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
    elif kind == "constant":
        x[:] = 5.0
    return x


def timed(fn, n=3):
    fn(); dev.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    dev.sync()
    return round((time.perf_counter() - t0) / n * 1e3, 3)


out = {}
kinds = ("kelvin", "celsius", "precip", "mask30", "nan10", "ties", "constant")
for T in (365, 930, 10950):
    ta = TimeAxis.daily("1981-01-01", T, "noleap")
    for kind in kinds:
        x = dev.to_device(field(kind, T))
        y = dev.to_device(field(kind, T))
        af = dev.to_device(rng.normal(0, 1, (20, C)).astype(np.float32))
        res = {"eqm_train": timed(lambda: K.eqm_train(dev, x, y, q, "+" if kind != "precip" else "*")),
               "qdm_nearest": timed(lambda: K.qdm_adjust(dev, x, af, q, "+", "nearest", "constant")),
               "qdm_linear": timed(lambda: K.qdm_adjust(dev, x, af, q, "+", "linear", "constant"))}
        if T != 930:
            res["percentile_doy"] = timed(lambda: xcal.percentile_doy(x, ta, window=5, per=90.0, device=dev))
        out[f"T{T}_{kind}"] = res
        del x, y, af
base = {T: out[f"T{T}_kelvin"] for T in (365, 930, 10950)}
worst = {}
for k, v in out.items():
    T = int(k.split("_")[0][1:])
    for op, ms in v.items():
        ratio = round(ms / base[T][op], 2)
        if ratio > 1.5:
            worst[f"{k}.{op}"] = [ms, base[T][op], ratio]
print(json.dumps({"ms": out, "slower_than_1.5x_kelvin": worst}))
