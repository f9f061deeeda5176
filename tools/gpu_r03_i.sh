#!/bin/bash
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r03i; rm -rf $O; mkdir -p $O
export XH_DIAGNOSTICS=1
cd /tmp && export TMPDIR=/tmp
for v in "XH_HIST_GEOM=32" "XH_HIST_GEOM=64" "XH_HIST_GEOM=128" "XH_HIST_GEOM=256"; do
  rm -rf $O/prof
  env $v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o c4 -- python $GRAFT_REPO_ROOT/tools/bench_c4.py > $O/prof.log 2>&1
  echo "== $v" >> $O/ab.log
  python $GRAFT_REPO_ROOT/tools/kstats.py $O/prof 4 | grep -i "stream_test" >> $O/ab.log
done
rm -rf $O/prof
cat $O/ab.log
