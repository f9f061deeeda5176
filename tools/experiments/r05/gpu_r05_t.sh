#!/bin/bash
# round 5, call T: plane fuzzer + durations of the slow sdba tests
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r05t; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
for s in 1 2 3 4; do FUZZ_SEED=$s timeout 200 python tools/fuzz_plane.py 40 > $O/fuzz_plane_$s.txt 2>&1; tail -1 $O/fuzz_plane_$s.txt | cut -c1-1500; done
