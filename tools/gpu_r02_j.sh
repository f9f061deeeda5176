#!/bin/bash
# k_select_lean with the LDS-DMA column prefetch: tests, config-4 train, phase timers
cd $GRAFT_REPO_ROOT
export XH_DIAGNOSTICS=1
O=gpurun_out/r02s; mkdir -p $O; rm -f $O/glds.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -q -x -k "quantile or eqm or select" 2>&1 | tail -5 | tee -a $O/glds.txt
for v in 0 1; do
  echo "XH_LEAN_NOGLDS=$v" | tee -a $O/glds.txt
  XH_LEAN_NOGLDS=$v timeout 300 python tools/bench_c4.py 2>&1 | tail -1 | tee -a $O/glds.txt
done
echo "XH_LEAN_NOGLDS=0 NT=256" | tee -a $O/glds.txt
XH_LEAN_NT=256 timeout 300 python tools/bench_c4.py 2>&1 | tail -1 | tee -a $O/glds.txt
XH_SELECT_PROF=1 timeout 300 python tools/bench_c4.py 121600 2>&1 | tail -13 | tee -a $O/glds.txt
