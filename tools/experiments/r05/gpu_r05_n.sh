#!/bin/bash
# round 5, call N: xh_plane_linear row kernel with per-workgroup work-list stretches: scipy comparisons, phase timing
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r05n; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest -m gpu -q --tb=short tests/test_gpu_plane.py tests/test_gpu_api.py -k "plane or sub_groupings or grouped" > $O/pytest.log 2>&1; echo "rc=$?" | tee $O/summary.txt
tail -5 $O/pytest.log
timeout 600 python - > $O/plane_time.txt 2>&1 <<'PY'
import os, sys, json
sys.path.insert(0, os.getcwd())
import bench
from xclim_amd import kernels as K
from xclim_amd._capi import Device
dev = Device(0)
os.environ["XH_DIAGNOSTICS"] = "1"
for abl in ("0", "1", "2"):
    os.environ["XH_PLANE_ABL"] = abl
    r = bench.bench_plane_linear(dev, K, 1440 * 90); print("XH_PLANE_ABL", abl, "ms", r["ms"])
PY
tail -4 $O/plane_time.txt | cut -c1-300
