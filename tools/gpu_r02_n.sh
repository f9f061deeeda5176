#!/bin/bash
cd $GRAFT_REPO_ROOT
export XH_DIAGNOSTICS=1
O=gpurun_out/r02ae; mkdir -p $O; rm -f $O/abl.txt
for a in 0 4 32 64; do
  XH_SELECT_ABL=$a timeout 200 python tools/bench_lean2.py 2>&1 | tail -1 | tee -a $O/abl.txt
done
