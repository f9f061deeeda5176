#!/bin/bash
# round 5, last call: every fuzzer on a fresh seed (45 s each), then the driver's bench command on the final tree
set -u
cd $GRAFT_REPO_ROOT
FUZZ_SEED=90531 SECS=45 bash tools/gpu_fuzz_all.sh
bash tools/gpu_bench_r05.sh
