#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03l; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_adapter.py tests/test_gpu_api.py tests/test_gpu_blocks.py -m gpu -q --tb=short -k "threshold_count or tx90p or tx10p" > $O/pytest_sel.log 2>&1; echo "sel tests rc=$?" | tee $O/summary.txt; tail -30 $O/pytest_sel.log
timeout 600 python tools/bench_tx30.py > $O/tx30.log 2>&1; tail -5 $O/tx30.log
XH_DIAGNOSTICS=1 XH_TCOUNT_LEGACY=1 timeout 600 python tools/bench_tx30.py > $O/tx30_legacy.log 2>&1; tail -3 $O/tx30_legacy.log
