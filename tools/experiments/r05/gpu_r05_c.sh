#!/bin/bash
# round 5, call C: the fused two-pass select (k_hs_fused) — bitwise fuzz, A/B timing at config 4, the selection tests
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r05c; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python tools/fuzz_r05.py 60 > $O/fuzz_r05.txt 2>&1; tail -2 $O/fuzz_r05.txt
export XH_DIAGNOSTICS=1
for rep in 1 2; do
for f in 0 1 2; do
  echo -n "XH_HIST_FUSED=$f " | tee -a $O/c4_ab.txt
  XH_HIST_FUSED=$f timeout 300 python tools/bench_c4.py 2>&1 | tail -1 | tee -a $O/c4_ab.txt
done
done
unset XH_DIAGNOSTICS
timeout 900 python -m pytest -m gpu -q --tb=short -x tests/test_gpu_fullsize.py tests/test_gpu_kernels.py -k "quantile or eqm or select or c4 or config4" > $O/pytest.log 2>&1; echo "rc=$?" | tee $O/summary.txt
tail -5 $O/pytest.log
