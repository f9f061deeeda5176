#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d; rm -rf $O; mkdir -p $O
export XH_DIAGNOSTICS=1 XH_HIST_STATS=1
cd /tmp && export TMPDIR=/tmp
for v in "XH_HIST_ABL=0" "XH_HIST_ABL=2" "XH_HIST_ABL=3"; do
  rm -rf $GRAFT_REPO_ROOT/$O/prof
  env $v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o c4 -- python $GRAFT_REPO_ROOT/tools/bench_c4.py > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
  echo "== $v" >> $GRAFT_REPO_ROOT/$O/ab.log
  python $GRAFT_REPO_ROOT/tools/kstats.py $GRAFT_REPO_ROOT/$O/prof 4 >> $GRAFT_REPO_ROOT/$O/ab.log
done
rm -rf $GRAFT_REPO_ROOT/$O/prof
cat $GRAFT_REPO_ROOT/$O/ab.log
