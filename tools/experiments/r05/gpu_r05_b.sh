#!/bin/bash
# round 5, call B: Infinity-Cache re-read, second pass in reverse row order; group barrier with relaxed polling
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r05b; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 tools/mall_ubench tile 10950 1036800 0 > $O/mall_tile_fwd.txt 2>&1
timeout 600 tools/mall_ubench tile 10950 1036800 1 > $O/mall_tile_rev.txt 2>&1
cat $O/mall_tile_fwd.txt $O/mall_tile_rev.txt
