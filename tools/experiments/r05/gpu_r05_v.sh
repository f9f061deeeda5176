#!/bin/bash
# round 5, call V: Grouper(add_dims)
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r05v; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest -m gpu -q --tb=short tests/test_gpu_api.py -k "add_dims or sub_groupings or qdm_grouped" --durations=5 > $O/pytest.log 2>&1; echo "rc=$?" | tee $O/summary.txt
tail -30 $O/pytest.log
