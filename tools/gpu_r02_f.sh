#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02f; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --tb=short -x -k "register_sort or one_year or quantile_series or eqm or coord" > $O/pytest_new.log 2>&1; echo "new tests rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest_new.log
for a in 0 8 12; do
  echo "abl=$a"; XH_DIAGNOSTICS=1 XH_REGSORT_ABL=$a timeout 300 python tools/bench_eqm.py 365 2>&1 | tail -1 | cut -c1-110
done | tee $O/abl3.txt
timeout 600 python tools/bench_c4.py | tee $O/c4.txt
XH_DIAGNOSTICS=1 XH_STREAM2_PRIO=1 timeout 600 python tools/bench_c4.py | tee -a $O/c4.txt
