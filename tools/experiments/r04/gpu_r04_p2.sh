#!/bin/bash
# round 4, percentile_doy on 30 years: k_pdoy_quad against k_pdoy_top16 (XH_PDOY_QUAD=0), chunk lengths, same box
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04p2; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest -m gpu -q --tb=short tests -k "doy or pdoy or tx90 or percentile or bootstrap or tn10 or golden" > $O/pytest.log 2>&1; echo "rc=$?" | tee $O/summary.txt
tail -15 $O/pytest.log
export XH_DIAGNOSTICS=1
for rep in 1 2; do
  XH_PDOY_QUAD=0 XH_PDOY_CHUNK=24 python tools/bench_pdoy30.py >> $O/res.jsonl 2>>$O/err.log
  for ch in ${CHUNKS:-24 46 92 184}; do
    XH_PDOY_CHUNK=$ch python tools/bench_pdoy30.py >> $O/res.jsonl 2>>$O/err.log
  done
  PERS=10 python tools/bench_pdoy30.py >> $O/res.jsonl 2>>$O/err.log
done
cat $O/res.jsonl; tail -3 $O/err.log
