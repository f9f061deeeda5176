#!/bin/bash
# round 5, call U: could the day-of-year training (365 groups x gather + select) be ONE percentile_doy call per field?
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r05u; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python - > $O/doy_train.txt 2>&1 <<'PY'
import os, sys, json, time
import numpy as np
sys.path.insert(0, os.getcwd())
import bench
from xclim_amd import kernels as K, sdba
from xclim_amd._capi import Device
from xclim_amd.timeaxis import TimeAxis
dev = Device(0)
T, C = 10950, 1440 * 90
ta = TimeAxis.daily("1981-01-01", T, "noleap")
base = bench.seasonal_base(T)
ref = K.fill_synthetic(dev, T, C, 0, 4, base, 3.0)
hist = K.fill_synthetic(dev, T, C, 0, 5, base + np.float32(1.5), 3.3)
def timed(label, fn, n=2):
    fn(); dev.sync()
    t0 = time.perf_counter()
    for _ in range(n): r = fn()
    dev.sync()
    print(label, round((time.perf_counter() - t0) / n * 1e3, 1), "ms", flush=True)
    return r
tb, years, doys = ta.doy_table()
q = sdba.equally_spaced_nodes(20)
for window in (31, 15, 5):
    os.environ.pop("XH_DIAGNOSTICS", None)
    tab = timed(f"percentile_doy window {window} x 20 nodes (one field)", lambda: K.percentile_doy(dev, ref, tb, window, q * 100.0, 1.0, 1.0), 1)
    eqm = timed(f"train time.dayofyear window {window} (two fields)", lambda: sdba.EmpiricalQuantileMapping.train(ref, hist, nquantiles=20, kind="+", group="time.dayofyear", window=window, time=ta, device=dev), 1)
    a = tab.get()            # (nq, ndoy, C) float64
    tab_h = K.percentile_doy(dev, hist, tb, window, q * 100.0, 1.0, 1.0).get()
    hq = eqm.hist_q.reshape(365, 20, C)
    print("  max |hist_q - percentile_doy|", float(np.nanmax(np.abs(hq - tab_h.transpose(1, 0, 2).astype(np.float32)))), "labels", eqm.group_labels[:3], doys[:3], flush=True)
PY
cat $O/doy_train.txt | tail -12
