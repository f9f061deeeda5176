#!/bin/bash
# every seeded fuzzer of tools/ on the GPU box, SECS seconds each
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/fuzz_all; rm -rf $O; mkdir -p $O
for f in fuzz_pdoy fuzz_pdoy_count fuzz_r03 fuzz_r04 fuzz_r05 fuzz_inf fuzz_regsort fuzz_f64 fuzz_plane fuzz_winsel fuzz_groups; do
  echo "== $f" | tee -a $O/log.txt
  timeout 900 python tools/$f.py ${SECS:-30} 2>&1 | tail -2 | tee -a $O/log.txt
done
