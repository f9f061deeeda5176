#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02r; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "transpose" 2>&1 | tail -3 | tee $O/t.txt
python tools/bench_kernels.py 2>&1 | grep "transpose" | tee -a $O/t.txt
XH_DIAGNOSTICS=1 XH_TRANSPOSE_64=1 python tools/bench_kernels.py 2>&1 | grep "transpose" | tee -a $O/t.txt
timeout 300 python tools/bench_c4.py 2>&1 | tail -1 | tee -a $O/t.txt
