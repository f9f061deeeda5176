#!/bin/bash
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r05w; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python tools/experiments/r05/dbg_add_dims.py > $O/dbg.txt 2>&1; tail -20 $O/dbg.txt
