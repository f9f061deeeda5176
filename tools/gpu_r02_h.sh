#!/bin/bash
# EQM C4 train: time-major in-place selection (no transposed scratch) vs the transposed-batch pipeline
cd $GRAFT_REPO_ROOT
export XH_DIAGNOSTICS=1
O=gpurun_out/r02q; mkdir -p $O; rm -f $O/tm.txt
for tm in 0 1 2; do
  echo "XH_SELECT_TM=$tm" | tee -a $O/tm.txt
  XH_SELECT_TM=$tm timeout 300 python tools/bench_c4.py 2>&1 | tail -1 | tee -a $O/tm.txt
done
XH_SELECT_TM=1 XH_SELECT_PROF=1 timeout 300 python tools/bench_c4.py 24320 2>&1 | tail -14 | tee -a $O/tm.txt
XH_SELECT_TM=1 timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -k "eqm_30yr" 2>&1 | tail -3 | tee -a $O/tm.txt
