#!/bin/bash
# round 5, call P: where does the time of a month-grouped EQM (the sdba notebook's configuration) go?  30 years x 1440 x 90
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r05p; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python - > $O/month_pipeline.txt 2>&1 <<'PY'
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
sim = K.fill_synthetic(dev, T, C, 0, 6, base + np.float32(3.5), 3.3)
def timed(label, fn, n=2):
    fn(); dev.sync()
    t0 = time.perf_counter()
    for _ in range(n): r = fn()
    dev.sync()
    print(label, round((time.perf_counter() - t0) / n * 1e3, 1), "ms", flush=True)
    return r
for group, window in (("time.month", 1), ("time.dayofyear", 31)):
    eqm = timed(f"train {group} window {window}", lambda: sdba.EmpiricalQuantileMapping.train(ref, hist, nquantiles=20, kind="+", group=group, window=window, time=ta, device=dev), 1)
    for interp in ("nearest", "linear"):
        timed(f"  adjust {interp}", lambda: eqm.adjust(sim, interp=interp, time=ta, keep=True), 2)
eqm = timed("train time", lambda: sdba.EmpiricalQuantileMapping.train(ref, hist, nquantiles=20, kind="+", device=dev), 2)
timed("  adjust nearest", lambda: eqm.adjust(sim, interp="nearest", keep=True), 2)
PY
cat $O/month_pipeline.txt | tail -12
