#!/bin/bash
# round 5, call A: Infinity-Cache re-read micro-benchmark (VERDICT r4 #1a) + this box's baseline bench line
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r05a; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 tools/mall_ubench flat > $O/mall_flat.txt 2>&1
timeout 600 tools/mall_ubench tile > $O/mall_tile.txt 2>&1
timeout 600 python bench.py --no-cpu > $O/bench.log 2>&1
tail -3 $O/mall_flat.txt; tail -30 $O/mall_tile.txt
