#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03e; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_adapter.py tests/test_gpu_patch.py -m gpu -q --tb=short > $O/pytest_adapter.log 2>&1; echo "adapter rc=$?" | tee $O/summary.txt
tail -60 $O/pytest_adapter.log
