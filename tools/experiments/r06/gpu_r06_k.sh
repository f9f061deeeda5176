#!/bin/bash
# round 6, call K: A/B of the EQM adjust kernel's prologue (node loads requested together) — two library builds alternating as
# processes on one box
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_k; rm -rf $O; mkdir -p $O
for i in 1 2 3; do
  for v in old new; do
    cp tools/experiments/r06/lib_$v.so xclim_amd/lib/libxclimhip.so
    echo "$v $(timeout 300 python tools/experiments/r06/eqm_adjust_time.py 2>/dev/null | tail -1)" | tee -a $O/ab.txt
  done
done
cp tools/experiments/r06/lib_new.so xclim_amd/lib/libxclimhip.so
timeout 900 python -m pytest tests -m gpu -q -x -k "eqm or adjust or dqm or qdm or plane or sub_group or interp" 2>&1 | tail -2
