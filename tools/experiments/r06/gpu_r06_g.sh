#!/bin/bash
# round 6, call G: A/B of two builds of the library on ONE box, alternating processes: EQM train at config 4 (tools/bench_c4.py)
# and QDM at config 4 (qdm_c4_abl.py 0).  lib_old.so / lib_new.so are built in the container before the call.
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_g; rm -rf $O; mkdir -p $O
for i in 1 2 3; do
  for v in old new; do
    cp tools/experiments/r06/lib_$v.so xclim_amd/lib/libxclimhip.so
    echo "$v $(timeout 300 python tools/bench_c4.py 2>/dev/null | tail -1)" | tee -a $O/ab_eqm.txt
    echo "$v $(timeout 300 python tools/experiments/r06/qdm_c4_abl.py 0 2>/dev/null | tail -1)" | tee -a $O/ab_qdm.txt
  done
done
cp tools/experiments/r06/lib_new.so xclim_amd/lib/libxclimhip.so
FUZZ_SEED=619 timeout 400 python tools/fuzz_r05.py 120 2>&1 | tail -1 | tee $O/fuzz_r05.txt
FUZZ_SEED=620 timeout 300 python tools/fuzz_r03.py 60 2>&1 | tail -1 | tee $O/fuzz_r03.txt
timeout 600 python -m pytest tests -m gpu -q -x -k "quantile_series or eqm or qdm or two_pass" 2>&1 | tail -2
