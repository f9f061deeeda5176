#!/bin/bash
# round 4, k_pdoy_walk (central percentiles on multi-year periods): tests, fuzz, timing against k_pdoy_merge (XH_PDOY_WALK=0)
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04w; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest -m gpu -q --tb=short -x tests -k "doy or pdoy or tx90 or percentile or bootstrap or tn10 or golden" > $O/pytest.log 2>&1; echo "rc=$?" | tee $O/summary.txt
tail -15 $O/pytest.log
timeout 300 python tools/fuzz_pdoy.py 40 2>&1 | tail -2
export XH_DIAGNOSTICS=1
for pers in 50 "25,50,75"; do
  PERS=$pers python tools/bench_pdoy30.py >> $O/res.jsonl 2>>$O/err.log
  PERS=$pers XH_PDOY_WALK=0 python tools/bench_pdoy30.py >> $O/res.jsonl 2>>$O/err.log
done
cat $O/res.jsonl; tail -3 $O/err.log
