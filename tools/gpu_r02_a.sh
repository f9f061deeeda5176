#!/bin/bash
# round-2 GPU session A: parity of the new kernels first, then the bench line and the EQM A/B
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --tb=short -x -k "register_sort or one_year or multi_year or quantile_series or eqm" > $O/pytest_new.log 2>&1; echo "new tests rc=$?" | tee -a $O/summary.txt
tail -15 $O/pytest_new.log
timeout 600 python tools/bench_eqm.py 365 > $O/eqm365_regsort.json 2> $O/eqm365_regsort.err; cat $O/eqm365_regsort.json
XH_DIAGNOSTICS=1 XH_SELECT_NOREGSORT=1 timeout 600 python tools/bench_eqm.py 365 > $O/eqm365_hist.json 2>&1; cat $O/eqm365_hist.json
XH_DIAGNOSTICS=1 XH_REGSORT_IRREGULAR=1 timeout 600 python tools/bench_eqm.py 365 > $O/eqm365_irregular.json 2>&1; cat $O/eqm365_irregular.json
timeout 1500 python -m pytest tests -m gpu -q --tb=short > $O/pytest_all.log 2>&1; echo "all tests rc=$?" | tee -a $O/summary.txt
tail -25 $O/pytest_all.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
cat $O/bench.json; tail -5 $O/bench.err
