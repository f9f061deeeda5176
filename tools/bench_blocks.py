"""PCIe-inclusive rate of the block adapter (xclim_amd/blocks.py) on the C2 workload: host float32 (365, 1440, 720) ->
tx90p counts on the host.  Compares pageable vs page-locked inputs and several slab widths with the one-shot
to_device -> compute -> get path.  Run on a GPU box: python tools/bench_blocks.py"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from xclim_amd import kernels as K
from xclim_amd._capi import get_device
from xclim_amd.blocks import map_cell_blocks
from xclim_amd.timeaxis import TimeAxis

dev = get_device()
T, Y, X = 365, 1440, 720
C = Y * X
ta = TimeAxis.daily("2001-01-01", T)
tb, years, doys = ta.doy_table()
seg, _ = ta.segments("YS")
period = np.zeros(len(doys), dtype=np.int32)

t0 = time.perf_counter()
pinned = dev.pinned_empty((T, Y, X), np.float32)
t_pin = time.perf_counter() - t0
rng = np.random.default_rng(0)
base = (288 + 12 * np.sin(2 * np.pi * (np.arange(T) - 100) / 365)).astype(np.float32)
for t in range(T):
    pinned[t] = base[t] + rng.standard_normal((Y, X), dtype=np.float32) * 3
pageable = np.array(pinned)


def chain(d, xs):
    cnt, _ = K.percentile_doy_count(d, xs, tb, 5, 90.0, ">", period, 1, want_valid=False)
    return cnt


def timed(f, n=3):
    f()
    best = 1e9
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        best = min(best, time.perf_counter() - t0)
    return best


def oneshot(x):
    d = dev.to_device(x.reshape(T, C))
    return chain(dev, d).get()


ref = oneshot(pageable)
rows = [{"case": "pinned_empty(1.5 GB)", "s": round(t_pin, 4)}]
for name, x in (("pageable", pageable), ("pinned", pinned)):
    s = timed(lambda: oneshot(x))
    rows.append({"case": f"one shot to_device+compute+get, {name}", "s": round(s, 4), "GB/s": round(x.nbytes / s / 1e9, 1)})
    for cb in (16384, 65536, 262144):
        out = map_cell_blocks(chain, [x], block_cells=cb, device=dev, pinned_out=(name == "pinned"))
        assert np.array_equal(out.reshape(1, C), ref)
        s = timed(lambda: map_cell_blocks(chain, [x], block_cells=cb, device=dev, pinned_out=(name == "pinned")))
        rows.append({"case": f"map_cell_blocks slab {cb}, {name}", "s": round(s, 4), "GB/s": round(x.nbytes / s / 1e9, 1)})
# EQM adjust: the output is as large as the input, so the copy-out lane matters (full-duplex PCIe)
q = (np.arange(20) + 0.5) / 20
dd = dev.to_device(pageable.reshape(T, C))
af_d, hq_d = K.eqm_train(dev, dd, dev.to_device(pageable.reshape(T, C) + np.float32(1.5)), q, "+")
af, hq = af_d.get().reshape(20, Y, X), hq_d.get().reshape(20, Y, X)
del dd


def adjust(d, s_, a_, h_):
    return K.eqm_adjust(d, s_, a_, h_, "+", "linear", "constant")


def adjust_oneshot(x):
    return adjust(dev, dev.to_device(x.reshape(T, C)), dev.to_device(af.reshape(20, C)), dev.to_device(hq.reshape(20, C))).get()


ref_scen = adjust_oneshot(pageable)
s = timed(lambda: adjust_oneshot(pageable))
rows.append({"case": "EQM adjust one shot, pageable", "s": round(s, 4), "GB/s in+out": round(2 * pageable.nbytes / s / 1e9, 1)})
afp, hqp = dev.pinned_empty(af.shape, np.float32), dev.pinned_empty(hq.shape, np.float32)
afp[...], hqp[...] = af, hq
outp, outg = dev.pinned_empty((T, Y, X), np.float32), np.empty((T, Y, X), np.float32)
for name, xs, o in (("pageable in / out", [pageable, af, hq], outg), ("pinned in / out", [pinned, afp, hqp], outp)):
    for cb in (32768, 65536, 262144):
        o[...] = 0
        map_cell_blocks(adjust, xs, block_cells=cb, device=dev, out=o)
        assert np.array_equal(o.reshape(T, C), ref_scen, equal_nan=True)
        s = timed(lambda: map_cell_blocks(adjust, xs, block_cells=cb, device=dev, out=o))
        rows.append({"case": f"EQM adjust map_cell_blocks slab {cb}, {name} (preallocated out)", "s": round(s, 4),
                     "GB/s in+out": round(2 * pageable.nbytes / s / 1e9, 1)})
d = dev.to_device(pageable.reshape(T, C))
s = timed(lambda: (chain(dev, d), dev.sync()))
rows.append({"case": "device resident (no PCIe)", "s": round(s, 5), "GB/s": round(pageable.nbytes / s / 1e9, 1)})
for r in rows:
    print(json.dumps(r))
