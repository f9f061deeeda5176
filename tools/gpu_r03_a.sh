#!/bin/bash
# round 3, call A: the two-pass histogram selection (select4.hip): parity tests, config-4 timing A/B, kernel stats, VALU ubench
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_edges.py tests/test_gpu_fuzz.py -m gpu -q --tb=short -k "quantile or eqm or select" > $O/pytest_sel.log 2>&1; echo "sel tests rc=$?" | tee $O/summary.txt
tail -5 $O/pytest_sel.log
export XH_DIAGNOSTICS=1 XH_HIST_STATS=1
timeout 300 python tools/bench_c4.py > $O/c4_hist.log 2>&1; tail -4 $O/c4_hist.log
XH_SELECT_NOHIST=1 timeout 300 python tools/bench_c4.py > $O/c4_legacy.log 2>&1; tail -2 $O/c4_legacy.log
XH_BENCH_T=3650 timeout 300 python tools/bench_c4.py > $O/c3650_hist.log 2>&1; tail -2 $O/c3650_hist.log
XH_BENCH_T=3650 XH_SELECT_NOHIST=1 timeout 300 python tools/bench_c4.py > $O/c3650_legacy.log 2>&1; tail -1 $O/c3650_legacy.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o c4 -- python $GRAFT_REPO_ROOT/tools/bench_c4.py > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_c4.csv
rm -rf $O/prof
head -12 $O/kernel_stats_c4.csv
timeout 120 tools/valu_ubench > $O/valu_ubench.txt 2>&1; grep -E "waves/SIMD 2 .*(CE|sub_co|cvt_u32|min_u32|add_u32)" $O/valu_ubench.txt
