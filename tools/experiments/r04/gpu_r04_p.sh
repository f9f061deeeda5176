#!/bin/bash
# round 4, pdoy_top float networks: the percentile_doy tests + the 30-year timing
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04p; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest -m gpu -q --tb=short tests -k "doy or pdoy or tx90 or percentile or bootstrap or tn10 or golden" > $O/pytest.log 2>&1; echo "rc=$?" | tee $O/summary.txt
tail -15 $O/pytest.log
timeout 600 python tools/bench_tx30.py > $O/tx30.json 2>$O/tx30.err; cat $O/tx30.json; tail -3 $O/tx30.err
