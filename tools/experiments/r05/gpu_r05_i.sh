#!/bin/bash
# round 5, call I: streaming QDM "nearest" (xh_qdm_hist): differential fuzz against the exact-rank kernels, oracle tests, timing
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r05i; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
FUZZ_ONLY=qdm timeout 400 python tools/fuzz_r05.py 90 > $O/fuzz_qdm.txt 2>&1; tail -3 $O/fuzz_qdm.txt
timeout 300 python tools/fuzz_r05.py 40 > $O/fuzz_all.txt 2>&1; tail -2 $O/fuzz_all.txt
timeout 900 python -m pytest -m gpu -q --tb=short -x tests/test_gpu_api.py tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -k "qdm or quantile or eqm or select" > $O/pytest.log 2>&1; echo "rc=$?" | tee $O/summary.txt
tail -5 $O/pytest.log
export XH_DIAGNOSTICS=1 XH_HIST_STATS=1
timeout 600 python - > $O/qdm_time.txt 2>&1 <<'PY'
import os, sys, json
import numpy as np
sys.path.insert(0, os.getcwd())
import bench
from xclim_amd import kernels as K
from xclim_amd._capi import Device
dev = Device(0)
q = (np.arange(20) + 0.5) / 20
for T, C in ((10950, 1440 * 720), (55152, 1440 * 90)):
    base = bench.seasonal_base(T)
    sim = K.fill_synthetic(dev, T, C, 0, 6, base + np.float32(3.5), 3.3)
    af = dev.to_device(np.random.default_rng(0).normal(1.5, 0.3, (20, C)).astype(np.float32))
    scen = dev.empty((T, C), np.float32)
    ms = bench.event_time(dev, lambda: K.qdm_adjust(dev, sim, af, q, "+", "nearest", "constant", out=scen), 2)
    print(json.dumps({"T": T, "C": C, "qdm_nearest_ms": ms, "frac": 8.0 * T * C / ms / 1e6 / 8000.0}))
    for a in (sim, af, scen): a.free()
PY
cat $O/qdm_time.txt | tail -8
