#!/bin/bash
# transposed-scratch column stride: HBM channel spread of the lockstep column streams
cd $GRAFT_REPO_ROOT
export XH_DIAGNOSTICS=1
O=gpurun_out/r02t; mkdir -p $O; rm -f $O/tpad.txt
for pad in 0 1 2 3 4 5 7 8 9 16 17; do
  echo "TPAD=$pad" | tee -a $O/tpad.txt
  XH_SELECT_TPAD=$pad timeout 300 python tools/bench_c4.py 2>&1 | tail -1 | tee -a $O/tpad.txt
done
echo "TPAD=1 NOGLDS" | tee -a $O/tpad.txt
XH_SELECT_TPAD=1 XH_LEAN_NOGLDS=1 timeout 300 python tools/bench_c4.py 2>&1 | tail -1 | tee -a $O/tpad.txt
echo "TPAD=1 NOGLDS NT256" | tee -a $O/tpad.txt
XH_SELECT_TPAD=1 XH_LEAN_NOGLDS=1 XH_LEAN_NT=256 timeout 300 python tools/bench_c4.py 2>&1 | tail -1 | tee -a $O/tpad.txt
