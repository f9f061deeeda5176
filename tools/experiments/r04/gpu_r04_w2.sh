#!/bin/bash
# round 4, k_pdoy_walk: walk steps per (wave, day, percentile), phases off (XH_PDOY_ABL 1 = no selection, 2 = no sort)
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04w2; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest -m gpu -q --tb=short -x tests/test_gpu_kernels.py -k "doy or percentile" > $O/pytest.log 2>&1; echo "rc=$?"; tail -3 $O/pytest.log
timeout 300 python tools/fuzz_pdoy.py 25 2>&1 | tail -1
export XH_DIAGNOSTICS=1
XH_PDOY_WALK_STEPS=1 PERS=50 python tools/bench_pdoy30.py 2>&1 | tail -2
for abl in 0 1; do XH_PDOY_ABL=$abl PERS=50 python tools/bench_pdoy30.py 2>&1 | tail -1; done
PERS=25,50,75 python tools/bench_pdoy30.py 2>&1 | tail -1
