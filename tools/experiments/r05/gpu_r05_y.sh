#!/bin/bash
# round 5, call Y: k_plane_pair timing after prefetch + lockstep searches; quick correctness
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r05y; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest -m gpu -q --tb=short tests/test_gpu_plane.py -x > $O/pytest.log 2>&1; echo "rc=$?" | tee $O/summary.txt
tail -3 $O/pytest.log
FUZZ_SEED=21 timeout 200 python tools/fuzz_plane.py 30 > $O/fuzz_plane.txt 2>&1; tail -1 $O/fuzz_plane.txt | cut -c1-1200
bash tools/experiments/r05/gpu_r05_p.sh 2>&1 | grep -A3 "train time.month"
XH_DIAGNOSTICS=1 XH_PLANE_ABL=1 bash tools/experiments/r05/gpu_r05_p.sh 2>&1 | grep -A3 "train time.month" 
