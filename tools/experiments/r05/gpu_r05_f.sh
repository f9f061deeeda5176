#!/bin/bash
# round 5, call F: adapter tests (device-resident inputs / tables, new replays) + the bench line
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r05f; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest -m gpu -q --tb=short tests/test_gpu_adapter.py tests/test_gpu_patch.py tests/test_gpu_blocks.py > $O/pytest.log 2>&1; echo "rc=$?" | tee $O/summary.txt
tail -40 $O/pytest.log
timeout 900 python bench.py --no-cpu > $O/bench.log 2>&1
tail -1 $O/bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d['extra']['adapter_e2e'], indent=1))"
