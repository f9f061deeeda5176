#!/bin/bash
cd $GRAFT_REPO_ROOT
export XH_DIAGNOSTICS=1
O=gpurun_out/r02ab; mkdir -p $O; rm -f $O/t.txt
for T in 3650 7300; do for nt in 256 512; do
  echo "T=$T NT=$nt" | tee -a $O/t.txt
  XH_BENCH_T=$T XH_LEAN_NT=$nt timeout 300 python tools/bench_c4.py 2>&1 | tail -1 | tee -a $O/t.txt
done; done
