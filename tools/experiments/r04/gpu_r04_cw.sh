#!/bin/bash
# (record of an experiment: the XH_HIST_CW switch left the tree with it, profiles/r04/select4_anatomy.txt #9)
# round 4, select4 with 32-column tiles (two 512-thread workgroups per CU, XH_HIST_CW=32) against 64-column tiles: the
# selection tests + fuzz under CW=32, then config-4 training time for both (same box)
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04cw; rm -rf $O; mkdir -p $O
export XH_DIAGNOSTICS=1
XH_HIST_CW=32 timeout 900 python -m pytest -m gpu -q --tb=short -x tests -k "histogram or quantile_series or beyond_32768 or value_classes or eqm or fullsize" > $O/pytest.log 2>&1; echo "rc=$?"; tail -6 $O/pytest.log
XH_HIST_CW=32 timeout 300 python tools/fuzz_r04.py 40 2>&1 | tail -1
for rep in 1 2; do
  for cw in 64 32; do
    XH_HIST_CW=$cw XH_HIST_STATS=1 python tools/bench_c4.py 2>&1 | tail -2 | tr '\n' ' '; echo " cw=$cw"
  done
done
