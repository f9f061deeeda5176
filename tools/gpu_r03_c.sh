#!/bin/bash
# round 3, call C: A/B of the XCD-aware tile map, workgroups per CU and the epilogue sort of select4.hip on config 4
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c; mkdir -p $O
export XH_DIAGNOSTICS=1 XH_HIST_STATS=1
for v in "" "XH_HIST_NOXCD=1" "XH_HIST_GRID=1" "XH_HIST_ABL=1" "XH_HIST_NOXCD=1 XH_HIST_ABL=1"; do
  echo "== $v" >> $O/ab.log
  env $v timeout 300 python tools/bench_c4.py 2>&1 | tail -2 >> $O/ab.log
done
cat $O/ab.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -k "two_pass or many_columns or hard_dist" > $O/pytest_sel.log 2>&1; echo "sel tests rc=$?" | tee $O/summary.txt; tail -3 $O/pytest_sel.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o c4 -- python $GRAFT_REPO_ROOT/tools/bench_c4.py > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_c4.csv
rm -rf $O/prof
head -5 $O/kernel_stats_c4.csv | cut -c1-60,300-420
