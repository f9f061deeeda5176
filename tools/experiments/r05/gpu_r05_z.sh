#!/bin/bash
# round 5, call Z: kernel-trace of the month-grouped linear adjust (k_plane_pair / k_plane_work shares)
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r05z; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cat > /tmp/month_lin.py <<'PY'
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
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
eqm = sdba.EmpiricalQuantileMapping.train(ref, hist, nquantiles=20, kind="+", group="time.month", time=ta, device=dev)
for _ in range(3):
    eqm.adjust(sim, interp="linear", time=ta, keep=True)
dev.sync()
PY
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python /tmp/month_lin.py > $O/log.txt 2>&1
python $GRAFT_REPO_ROOT/tools/kstats.py $O/stats 12 | tee $O/kstats.txt
find $O -name "*.csv" -size +5M -delete
