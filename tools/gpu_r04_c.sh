#!/bin/bash
# Round 4, call C: loads in flight.  Ring of 2 / 3 / 4 register sets in the streaming loop of select4.hip, bare (abl 66 = both
# passes loads only) and complete.
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c; rm -rf $O; mkdir -p $O
export XH_DIAGNOSTICS=1
cd /tmp && export TMPDIR=/tmp
for ns in 2 3 4; do for abl in 66 0; do
  XH_HIST_NSET=$ns XH_HIST_ABL=$abl timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/n${ns}a$abl -o s -- python $GRAFT_REPO_ROOT/tools/bench_c4.py > $GRAFT_REPO_ROOT/$O/n${ns}a$abl.log 2>&1
  echo "nset=$ns abl=$abl $(python $GRAFT_REPO_ROOT/tools/kstats.py $GRAFT_REPO_ROOT/$O/n${ns}a$abl 4 | grep -E 'k_hs_hist|k_hs_collect' | awk '{print $1, $(NF-1)}' | tr '\n' ' ') $(tail -1 $GRAFT_REPO_ROOT/$O/n${ns}a$abl.log | cut -c1-60)" | tee -a $GRAFT_REPO_ROOT/$O/summary.txt
  find $GRAFT_REPO_ROOT/$O/n${ns}a$abl -type f ! -name "*kernel_stats.csv" -delete
done; done
unset XH_DIAGNOSTICS
cd $GRAFT_REPO_ROOT; timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -x -k "quantile_series" 2>&1 | tail -3
