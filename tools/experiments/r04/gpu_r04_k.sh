#!/bin/bash
# Round 4, call K: early priming (next tile's first batches requested before the epilogue) on / off, same box, alternating.
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04l; rm -rf $O; mkdir -p $O
export XH_DIAGNOSTICS=1
cd /tmp && export TMPDIR=/tmp
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/$tag -o s -- python $GRAFT_REPO_ROOT/tools/bench_c4.py > $GRAFT_REPO_ROOT/$O/$tag.log 2>&1
  echo "$tag: $(python $GRAFT_REPO_ROOT/tools/kstats.py $GRAFT_REPO_ROOT/$O/$tag 5 | grep -E 'k_hs_' | grep -v sample | awk '{n=$1; if (n=="void") n=$2" "$3; print substr(n,20,24), $(NF-1)}' | tr '\n' '|') $(grep -h train_ms $GRAFT_REPO_ROOT/$O/$tag.log | cut -c27-50)" | tee -a $GRAFT_REPO_ROOT/$O/summary.txt
  find $GRAFT_REPO_ROOT/$O/$tag -type f ! -name "*kernel_stats.csv" -delete
}
for i in 1 2; do
run e00_$i XH_HIST_EARLY=00
run e10_$i XH_HIST_EARLY=10
run e01_$i XH_HIST_EARLY=01
run e11_$i XH_HIST_EARLY=11
done
