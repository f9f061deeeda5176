#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02d; mkdir -p $O
for a in 0 8 9 11 15; do
  echo "abl=$a"; XH_DIAGNOSTICS=1 XH_REGSORT_ABL=$a timeout 300 python tools/bench_eqm.py 365 2>&1 | tail -1 | cut -c1-110
done | tee $O/abl2.txt
echo "small grid (L2/MALL resident): C=65536"; timeout 300 python tools/bench_eqm.py 365 65536 2>&1 | tail -1 | cut -c1-110 | tee -a $O/abl2.txt
