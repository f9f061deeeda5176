#!/bin/bash
# round 4, k_pdoy_quad anatomy: phases switched off (XH_PDOY_ABL: 1 = no sorting networks, 2 = no merges / selection)
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04p4; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest -m gpu -q --tb=short tests -k "doy or pdoy or tx90 or percentile or bootstrap or tn10 or golden" > $O/pytest.log 2>&1; echo "rc=$?" | tee $O/summary.txt
tail -5 $O/pytest.log
export XH_DIAGNOSTICS=1
for rep in 1 2; do
  for abl in 0 1 2 3; do
    XH_PDOY_ABL=$abl python tools/bench_pdoy30.py >> $O/res.jsonl 2>>$O/err.log
  done
  for ch in 24 46 184 366; do
    XH_PDOY_CHUNK=$ch python tools/bench_pdoy30.py >> $O/res.jsonl 2>>$O/err.log
  done
  XH_PDOY_QUAD=0 python tools/bench_pdoy30.py >> $O/res.jsonl 2>>$O/err.log
done
cat $O/res.jsonl; tail -3 $O/err.log
