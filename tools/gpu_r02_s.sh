#!/bin/bash
cd $GRAFT_REPO_ROOT
export XH_DIAGNOSTICS=1
O=gpurun_out/r02am; mkdir -p $O; rm -f $O/t.txt
for k in 0 2 3 4 5 8; do
echo "XH_PIPE_SPLIT=$k" | tee -a $O/t.txt
XH_PIPE_SPLIT=$k timeout 300 python tools/bench_c4.py 2>&1 | tail -1 | tee -a $O/t.txt
done
