#!/bin/bash
# round 5, call J: the whole GPU suite + smoke, every fuzzer (20 s each), the driver's bench command
set -u
cd $GRAFT_REPO_ROOT
bash tools/gpu_tests_all.sh r05_all2
SECS=20 bash tools/gpu_fuzz_all.sh
bash tools/gpu_bench_r05.sh
