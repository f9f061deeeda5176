#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/qdm; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_api.py -m gpu -q --tb=short -x -k "qdm" > $O/pytest_sel.log 2>&1; echo "sel tests rc=$?" | tee $O/summary.txt; tail -40 $O/pytest_sel.log
timeout 300 python tools/bench_qdm.py > $O/qdm.log 2>&1; tail -5 $O/qdm.log
