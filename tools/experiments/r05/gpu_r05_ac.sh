#!/bin/bash
# round 5: the bootstrap extra of bench.py alone (timing + sanity)
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r05ac; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python - > $O/boot.txt 2>&1 <<'PY'
import json, sys, os
sys.path.insert(0, os.getcwd())
import bench
from xclim_amd import kernels as K
from xclim_amd._capi import get_device
dev = get_device()
print(json.dumps(bench.bench_bootstrap(dev, K, 1440 * 90)))
print(json.dumps(bench.bench_plane_month(dev, K, 1440 * 90)))
PY
tail -3 $O/boot.txt | cut -c1-700
