#!/bin/bash
# round 5, call Q: xh_plane_nearest (grouped "nearest" over the whole series in one call): oracle tests + pipeline timing
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r05q; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest -m gpu -q --tb=short tests/test_gpu_plane.py tests/test_gpu_api.py tests/test_gpu_adapter.py -k "plane or sub_groupings or grouped or adapt or dqm or xsdba" > $O/pytest.log 2>&1; echo "rc=$?" | tee $O/summary.txt
tail -5 $O/pytest.log
timeout 300 python tools/fuzz_r04.py 30 > $O/fuzz_r04.txt 2>&1; tail -1 $O/fuzz_r04.txt
bash tools/experiments/r05/gpu_r05_p.sh
