#!/bin/bash
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/qdmprof; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o q -- python $GRAFT_REPO_ROOT/tools/bench_qdm.py > $O/prof.log 2>&1
python $GRAFT_REPO_ROOT/tools/kstats.py $O/prof 8
rm -rf $O/prof
