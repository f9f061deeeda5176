#!/bin/bash
# Round 4, call A: the rewritten select4.hip (value-class bins, bit-pair table, masked appends) and the radix fallback of
# select5.hip: parity tests first, then config-4 train timings per append mode and the per-kernel view.
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04a; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -x -k "quantile_series or eqm" > $O/pytest_select.log 2>&1; echo "select tests rc=$?" | tee $O/summary.txt
tail -15 $O/pytest_select.log
export XH_DIAGNOSTICS=1
for ap in 2 1 0; do
  XH_HIST_APPEND=$ap XH_HIST_STATS=1 timeout 300 python tools/bench_c4.py > $O/c4_append$ap.log 2>&1; echo "append=$ap: $(tail -1 $O/c4_append$ap.log)" | tee -a $O/summary.txt
done
XH_HIST_ABL=1 timeout 300 python tools/bench_c4.py > $O/c4_nosort.log 2>&1; echo "no sort: $(tail -1 $O/c4_nosort.log)" | tee -a $O/summary.txt
unset XH_DIAGNOSTICS
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -o stats -- python $GRAFT_REPO_ROOT/tools/bench_c4.py > $GRAFT_REPO_ROOT/$O/stats.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/kstats.py $O/stats 8 | tee -a $O/summary.txt
find $O -name "*.csv" -size +5M -delete
find $O/stats -type f ! -name "*stats*.csv" -delete 2>/dev/null
timeout 600 python tools/fuzz_r03.py 60 > $O/fuzz_r03.log 2>&1; echo "fuzz rc=$?" | tee -a $O/summary.txt; tail -5 $O/fuzz_r03.log
