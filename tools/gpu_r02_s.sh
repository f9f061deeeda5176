#!/bin/bash
cd $GRAFT_REPO_ROOT
export XH_DIAGNOSTICS=1
O=gpurun_out/r02aj; mkdir -p $O; rm -f $O/t.txt
for g in 16 4 3 2; do for pr in 0 1; do
echo "GRID_PER_CU=$g PRIO=$pr" | tee -a $O/t.txt
XH_LEAN_GRID_PER_CU=$g XH_STREAM2_PRIO=$pr timeout 300 python tools/bench_c4.py 2>&1 | tail -1 | tee -a $O/t.txt
done; done
