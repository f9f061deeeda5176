#!/bin/bash
# round 5, call X: k_plane_pair (month groupings, fractional coordinates): tests, fuzzer, pipeline timing
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r05x; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest -m gpu -q --tb=short tests/test_gpu_plane.py tests/test_gpu_api.py tests/test_gpu_sdba_golden.py -k "plane or sub_groupings or grouped or dqm" > $O/pytest.log 2>&1; echo "rc=$?" | tee $O/summary.txt
tail -8 $O/pytest.log
for s in 11 12; do FUZZ_SEED=$s timeout 200 python tools/fuzz_plane.py 40 > $O/fuzz_plane_$s.txt 2>&1; tail -1 $O/fuzz_plane_$s.txt | cut -c1-1200; done
bash tools/experiments/r05/gpu_r05_p.sh 2>&1 | tail -9
XH_DIAGNOSTICS=1 XH_PLANE_ABL=1 bash tools/experiments/r05/gpu_r05_p.sh 2>&1 | grep -A2 "train time.month" 
