#!/bin/bash
# Round 4, call Q: multi-round collection (long series): tests + timings at T = 10950 (unchanged?) and T = 55152
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04q; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -x -k "quantile_series or eqm" 2>&1 | tail -3 | tee $O/pytest.txt
export XH_DIAGNOSTICS=1 XH_HIST_STATS=1
timeout 300 python tools/bench_c4.py 2>&1 | tail -2 | cut -c1-200 | tee -a $O/summary.txt
XH_BENCH_T=55152 timeout 300 python tools/bench_c4.py 129600 2>&1 | tail -3 | cut -c1-200 | tee -a $O/summary.txt
XH_BENCH_T=32000 timeout 300 python tools/bench_c4.py 259200 2>&1 | tail -3 | cut -c1-200 | tee -a $O/summary.txt
unset XH_DIAGNOSTICS
timeout 600 python tools/fuzz_r03.py 45 2>&1 | tail -2 | tee -a $O/summary.txt
