#!/bin/bash
# Round 4, call F: collect kernel without spill reloads in the streaming loop (hits in one register): same-box phases.
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04f; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -x -k "quantile_series" 2>&1 | tail -3 | tee $O/pytest.txt
export XH_DIAGNOSTICS=1
cd /tmp && export TMPDIR=/tmp
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/$tag -o s -- python $GRAFT_REPO_ROOT/tools/bench_c4.py > $GRAFT_REPO_ROOT/$O/$tag.log 2>&1
  echo "$tag: $(python $GRAFT_REPO_ROOT/tools/kstats.py $GRAFT_REPO_ROOT/$O/$tag 5 | grep -E 'k_hs_' | grep -v sample | awk '{n=$1; if (n=="void") n=$2" "$3; print substr(n,1,40), $(NF-1)}' | tr '\n' '|') $(grep -h train_ms $GRAFT_REPO_ROOT/$O/$tag.log | cut -c27-50)" | tee -a $GRAFT_REPO_ROOT/$O/summary.txt
  find $GRAFT_REPO_ROOT/$O/$tag -type f ! -name "*kernel_stats.csv" -delete
}
run bare XH_HIST_GEOM=16,2,163000,1 XH_HIST_ABL=66
run full XH_HIST_ABL=0
run nosort XH_HIST_ABL=1
run noappend XH_HIST_ABL=32
run loads_noepi XH_HIST_ABL=70
run full_n3 XH_HIST_ABL=0 XH_HIST_NSET=3
run full2 XH_HIST_ABL=0
