#!/bin/bash
# round 5, call E: xh_plane_linear against scipy griddata; the grouped EQM / QDM / DQM tests
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r05e; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest -m gpu -q --tb=short tests/test_gpu_plane.py tests/test_gpu_api.py -k "plane or sub_groupings or grouped or qdm or dqm" > $O/pytest.log 2>&1; echo "rc=$?" | tee $O/summary.txt
tail -60 $O/pytest.log
