#!/bin/bash
cd $GRAFT_REPO_ROOT
export XH_DIAGNOSTICS=1
O=gpurun_out/r02al; mkdir -p $O; rm -f $O/t.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_api.py -q -x -k "quantile or eqm or select or sdba" 2>&1 | tail -5 | tee -a $O/t.txt
timeout 300 python tools/bench_c4.py 2>&1 | tail -1 | tee -a $O/t.txt
XH_SELECT_NOKEYS=1 timeout 300 python tools/bench_c4.py 2>&1 | tail -1 | tee -a $O/t.txt
XH_BENCH_T=3650 timeout 300 python tools/bench_c4.py 2>&1 | tail -1 | tee -a $O/t.txt
XH_BENCH_T=3650 XH_SELECT_NOKEYS=1 timeout 300 python tools/bench_c4.py 2>&1 | tail -1 | tee -a $O/t.txt
