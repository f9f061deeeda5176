#!/bin/bash
cd $GRAFT_REPO_ROOT
export XH_DIAGNOSTICS=1
O=gpurun_out/r02y; mkdir -p $O; rm -f $O/ab.txt
for nt in 512 256; do for t64 in 0 1; do
  echo "NOGLDS NT=$nt TRANSPOSE_64=$t64" | tee -a $O/ab.txt
  XH_LEAN_NOGLDS=1 XH_LEAN_NT=$nt XH_TRANSPOSE_64=$t64 timeout 300 python tools/bench_c4.py 2>&1 | tail -1 | tee -a $O/ab.txt
done; done
