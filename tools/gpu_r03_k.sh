#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03k; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -m gpu -q --tb=short -k "quantile or eqm" > $O/pytest_sel.log 2>&1; echo "sel tests rc=$?" | tee $O/summary.txt; tail -4 $O/pytest_sel.log
export XH_DIAGNOSTICS=1 XH_HIST_STATS=1
cd /tmp && export TMPDIR=/tmp
for v in "XH_HIST_DEFER=0"; do
  rm -rf $O/prof
  env $v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o c4 -- python $GRAFT_REPO_ROOT/tools/bench_c4.py > $O/prof.log 2>&1
  echo "== $v" >> $O/ab.log; grep train_ms $O/prof.log >> $O/ab.log
  python $GRAFT_REPO_ROOT/tools/kstats.py $O/prof 4 >> $O/ab.log
done
rm -rf $O/prof
cat $O/ab.log
