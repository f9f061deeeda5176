#!/bin/bash
# round 5, call R: k_qdm_regsort with the boundary ranks from a table and the run rule for ties inside the pick phase
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r05r; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest -m gpu -q --tb=short tests/test_gpu_api.py -k "qdm" > $O/pytest.log 2>&1; echo "rc=$?" | tee $O/summary.txt
tail -5 $O/pytest.log
timeout 300 python tools/fuzz_r03.py 60 > $O/fuzz_r03.txt 2>&1; tail -1 $O/fuzz_r03.txt
XH_QDM_STATS=1 XH_DIAGNOSTICS=1 timeout 300 python tools/debug_qdm.py > $O/debug_qdm.txt 2>&1; tail -8 $O/debug_qdm.txt
for env in "" "XH_QDM_LUT_GLOBAL=1" "XH_QDM_NOLUT=1 XH_QDM_NORUN=1"; do
  echo "== $env" | tee -a $O/abl.txt
  env $env XH_DIAGNOSTICS=1 timeout 300 python tools/bench_qdm_abl.py 0 32 33 34 35 36 48 0 2>&1 | tail -1 | tee -a $O/abl.txt
done
