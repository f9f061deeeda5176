#!/bin/bash
# round 4, k_pdoy_quad: sensitivity to the waves per CU (unused dynamic LDS caps them: 160 KB / pad)
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04p6; rm -rf $O; mkdir -p $O
export XH_DIAGNOSTICS=1
for rep in 1 2; do
  for pad in 0 16384 20480 27000 40000; do
    XH_PDOY_LDSPAD=$pad python tools/bench_pdoy30.py >> $O/res.jsonl 2>>$O/err.log
  done
done
cat $O/res.jsonl; tail -3 $O/err.log
