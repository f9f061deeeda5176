#!/bin/bash
# HBM traffic (PMC FETCH_SIZE / WRITE_SIZE, separate passes) of the secondary kernels exercised by tools/bench_kernels.py.
# Usage (GPU box): tools/profile_kernels.sh <tag>   ->   gpurun_out/pmck_<tag>/
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmck_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/tools/bench_kernels.py"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- $CMD > $OUT/pmc_write.log 2>&1
ls $OUT/*/ | head
