#!/bin/bash
# round 5: day-of-year training through the ring of windowed samples: tests + timing
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r05ab; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest -m gpu -q --tb=short tests/test_gpu_api.py tests/test_gpu_sdba_golden.py -k "sub_groupings or grouped or dqm or add_dims or adapt" > $O/pytest.log 2>&1; echo "rc=$?" | tee $O/summary.txt
tail -4 $O/pytest.log
bash tools/experiments/r05/gpu_r05_p.sh 2>&1 | tail -9
