#!/bin/bash
# round 5, call S: DQM with a windowed sub-grouping (xh_window_nanmean) + QDM one-year timing on both field kinds
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r05s; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest -m gpu -q --tb=short tests/test_gpu_api.py -k "dqm or qdm" > $O/pytest.log 2>&1; echo "rc=$?" | tee $O/summary.txt
tail -15 $O/pytest.log
timeout 300 python tools/bench_qdm.py > $O/bench_qdm.txt 2>&1; tail -3 $O/bench_qdm.txt
