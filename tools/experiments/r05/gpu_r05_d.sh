#!/bin/bash
# round 5, call D: in-process A/B of the fused select at config 4
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r05d; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python tools/bench_c4_ab.py 1036800 8 > $O/c4_ab.txt 2>&1; cat $O/c4_ab.txt
