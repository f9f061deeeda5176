#!/bin/bash
# EQM C4 train: scratch batch size sweep (does the transposed batch stay in the 256 MB Infinity Cache?)
cd $GRAFT_REPO_ROOT
export XH_DIAGNOSTICS=1
O=gpurun_out/r02p; mkdir -p $O; rm -f $O/batch.txt
for mb in 512 256 128 96 64 48 32 16; do
  echo "batch_mb=$mb" | tee -a $O/batch.txt
  XH_SELECT_BATCH_MB=$mb timeout 300 python tools/bench_c4.py 2>&1 | tail -1 | tee -a $O/batch.txt
done
for mb in 64 512; do
echo "batch_mb=$mb prio" | tee -a $O/batch.txt
XH_SELECT_BATCH_MB=$mb XH_STREAM2_PRIO=1 timeout 300 python tools/bench_c4.py 2>&1 | tail -1 | tee -a $O/batch.txt
echo "batch_mb=$mb cumask 4" | tee -a $O/batch.txt
XH_SELECT_BATCH_MB=$mb XH_STREAM2_CUMASK_STRIDE=4 timeout 300 python tools/bench_c4.py 2>&1 | tail -1 | tee -a $O/batch.txt
done
