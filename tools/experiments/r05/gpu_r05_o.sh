#!/bin/bash
# round 5, call O: candidate lists of 2049 .. 8192 keys sorted in LDS (no detour through the radix select / the global sort)
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r05o; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest -m gpu -q --tb=short -x tests/test_gpu_kernels.py tests/test_gpu_api.py tests/test_gpu_fullsize.py -k "quantile or eqm or qdm or select or 32768" > $O/pytest.log 2>&1; echo "rc=$?" | tee $O/summary.txt
tail -4 $O/pytest.log
FUZZ_ONLY=qdm timeout 300 python tools/fuzz_r05.py 60 > $O/fuzz_qdm.txt 2>&1; tail -1 $O/fuzz_qdm.txt
timeout 300 python tools/fuzz_r05.py 40 > $O/fuzz_all.txt 2>&1; tail -1 $O/fuzz_all.txt
timeout 300 python tools/fuzz_r03.py 30 > $O/fuzz_r03.txt 2>&1; tail -1 $O/fuzz_r03.txt
export XH_DIAGNOSTICS=1 XH_HIST_STATS=1
timeout 600 python - > $O/time55k.txt 2>&1 <<'PY'
import os, sys, json
import numpy as np
sys.path.insert(0, os.getcwd())
import bench
from xclim_amd import kernels as K
from xclim_amd._capi import Device
dev = Device(0)
q = (np.arange(20) + 0.5) / 20
T, C = 55152, 1440 * 90
base = bench.seasonal_base(T)
ref = K.fill_synthetic(dev, T, C, 0, 4, base, 3.0)
hist = K.fill_synthetic(dev, T, C, 0, 5, base + np.float32(1.5), 3.3)
af, hq = dev.empty((20, C), np.float32), dev.empty((20, C), np.float32)
ms = bench.event_time(dev, lambda: K.eqm_train(dev, ref, hist, q, "+", out=(af, hq)), 2)
print(json.dumps({"eqm_55k_train_ms": ms}))
scen = dev.empty((T, C), np.float32)
ms = bench.event_time(dev, lambda: K.qdm_adjust(dev, hist, af, q, "+", "nearest", "constant", out=scen), 2)
print(json.dumps({"qdm_55k_ms": ms}))
PY
tail -8 $O/time55k.txt | cut -c1-200
