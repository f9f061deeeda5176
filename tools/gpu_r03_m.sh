#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03m; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --tb=short -k "percentile_doy or pdoy or tx90p or tx10p or bootstrap or threshold_count or 30yr" > $O/pytest_sel.log 2>&1; echo "sel tests rc=$?" | tee $O/summary.txt; tail -30 $O/pytest_sel.log
timeout 600 python tools/bench_tx30.py > $O/tx30.log 2>&1; tail -5 $O/tx30.log
XH_DIAGNOSTICS=1 XH_PDOY_COUNT_FUSED=1 timeout 600 python tools/bench_tx30.py > $O/tx30_fusedkernel.log 2>&1; tail -2 $O/tx30_fusedkernel.log
XH_DIAGNOSTICS=1 XH_PDOY_COUNT_FUSED=1 timeout 900 python -m pytest tests -m gpu -q --tb=short -k "percentile_doy_count or tx90p or 30yr or fused" > $O/pytest_fusedkernel.log 2>&1; tail -3 $O/pytest_fusedkernel.log
