#!/bin/bash
cd $GRAFT_REPO_ROOT
export XH_DIAGNOSTICS=1
O=gpurun_out/r02x; mkdir -p $O; rm -f $O/abl.txt
for g in 0 1; do for a in 0 16 1 2 4 8 15 31; do
  XH_LEAN_NOGLDS=$g XH_SELECT_ABL=$a timeout 200 python tools/bench_lean2.py 2>&1 | tail -1 | tee -a $O/abl.txt
done; done
