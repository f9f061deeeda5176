#!/bin/bash
# round 5, call M: xh_plane_linear with the row kernel + work list: scipy comparisons, timing (doy), month coordinate timing
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r05m; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest -m gpu -q --tb=short tests/test_gpu_plane.py tests/test_gpu_api.py -k "plane or sub_groupings or grouped" > $O/pytest.log 2>&1; echo "rc=$?" | tee $O/summary.txt
tail -5 $O/pytest.log
timeout 600 python - > $O/plane_time.txt 2>&1 <<'PY'
import os, sys, json
import numpy as np
sys.path.insert(0, os.getcwd())
import bench
from xclim_amd import kernels as K
from xclim_amd._capi import Device
from xclim_amd.timeaxis import TimeAxis
from xclim_amd.sdba import Grouper
dev = Device(0)
r = bench.bench_plane_linear(dev, K, 1440 * 90); print("doy rows + work list:", r["ms"], r["frac"])
# month grouping, fractional coordinate, same band
T, G, nq, Cb = 10950, 12, 20, 1440 * 90
ta = TimeAxis.daily("1981-01-01", T, "noleap")
g = Grouper("time.month").coordinate(ta, True)
rng = np.random.default_rng(5)
node = (288.0 + 12.0 * np.sin(2 * np.pi * (np.arange(G) * 30.4 + 15 - 100) / 365))[:, None] + 3.0 * np.sort(rng.normal(0, 1, (G, nq)), axis=1)
hq = (node[:, :, None] + rng.normal(0, 0.2, Cb)[None, None, :]).astype(np.float32)
af = (1.5 + 0.3 * rng.normal(0, 1, (G, nq)))[:, :, None].astype(np.float32) + np.zeros((1, 1, Cb), np.float32)
d_hq, d_af = dev.to_device(hq), dev.to_device(af)
sim = K.fill_synthetic(dev, T, Cb, 0, 6, bench.seasonal_base(T), 3.3)
scen = dev.empty((T, Cb), np.float32)
gd = dev.to_device(g)
ms = bench.event_time(dev, lambda: K.plane_linear(dev, sim, gd, d_af, xq_all=d_hq, kind="+", out=scen), 1)
print("month (fractional coordinate, generic walk):", ms, "ms for", T * Cb / 1e9, "G queries")
PY
tail -3 $O/plane_time.txt | cut -c1-300
