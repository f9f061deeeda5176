#!/bin/bash
cd $GRAFT_REPO_ROOT
export XH_DIAGNOSTICS=1
O=gpurun_out/r02af; mkdir -p $O; rm -f $O/t.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_api.py -q -x -k "quantile or eqm or select or sdba" 2>&1 | tail -5 | tee -a $O/t.txt
for a in 0 4 2; do XH_SELECT_ABL=$a timeout 200 python tools/bench_lean2.py 2>&1 | tail -1 | tee -a $O/t.txt; done
timeout 300 python tools/bench_c4.py 2>&1 | tail -1 | tee -a $O/t.txt
python tools/bench_eqm_pr.py 2>&1 | tail -6 | tee -a $O/t.txt
