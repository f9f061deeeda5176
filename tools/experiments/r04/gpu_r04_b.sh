#!/bin/bash
# Round 4, call B: where do the two streaming passes of select4.hip spend their time?  Phase ablations (wrong results).
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b; rm -rf $O; mkdir -p $O
export XH_DIAGNOSTICS=1
cd /tmp && export TMPDIR=/tmp
for abl in 0 2 4 8 16 1 32 64 66; do
  XH_HIST_ABL=$abl timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/abl$abl -o s -- python $GRAFT_REPO_ROOT/tools/bench_c4.py > $GRAFT_REPO_ROOT/$O/abl$abl.log 2>&1
  echo "abl=$abl $(python $GRAFT_REPO_ROOT/tools/kstats.py $GRAFT_REPO_ROOT/$O/abl$abl 4 | grep -E 'k_hs_hist|k_hs_collect' | awk '{print $1, $(NF-1)}' | tr '\n' ' ')" | tee -a $GRAFT_REPO_ROOT/$O/summary.txt
  find $GRAFT_REPO_ROOT/$O/abl$abl -type f ! -name "*kernel_stats.csv" -delete
done
