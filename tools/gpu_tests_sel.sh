#!/bin/bash
# run a pytest selection on the GPU box: gpu_tests_sel.sh <tag> <pytest args...>
set -u
cd $GRAFT_REPO_ROOT
tag=$1; shift
O=gpurun_out/$tag; rm -rf $O; mkdir -p $O
timeout 1800 python -m pytest -m gpu -q --tb=short "$@" > $O/pytest.log 2>&1; echo "rc=$?" | tee $O/summary.txt
tail -40 $O/pytest.log
