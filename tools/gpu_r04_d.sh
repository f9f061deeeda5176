#!/bin/bash
# Round 4, call D: the bare streaming loop of select4.hip at different occupancies (row lanes, register sets, LDS bytes,
# workgroups per CU), one process, same box.
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d; rm -rf $O; mkdir -p $O
export XH_DIAGNOSTICS=1 XH_HIST_ABL=66
cd /tmp && export TMPDIR=/tmp
for g in 16,2,163000,1 16,2,0,2 16,3,0,2 8,2,81000,2 8,2,0,4 8,3,0,4 8,2,40000,4 4,2,0,8 4,4,0,8 4,2,40000,4; do
  XH_HIST_GEOM=$g timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/g$g -o s -- python $GRAFT_REPO_ROOT/tools/bench_c4.py > $GRAFT_REPO_ROOT/$O/g$g.log 2>&1
  echo "geom=$g $(python $GRAFT_REPO_ROOT/tools/kstats.py $GRAFT_REPO_ROOT/$O/g$g 5 | grep -E 'k_hs_stream_test' | awk '{print $(NF-1)}')" | tee -a $GRAFT_REPO_ROOT/$O/summary.txt
  find $GRAFT_REPO_ROOT/$O/g$g -type f ! -name "*kernel_stats.csv" -delete
done
