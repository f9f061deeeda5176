#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fuzz_inf
for seed in ${SEEDS:-777 778}; do
  FUZZ_SEED=$seed timeout 600 python tools/fuzz_inf.py ${SECS:-40} 2>&1 | tail -3 | tee -a gpurun_out/fuzz_inf/log.txt
done
