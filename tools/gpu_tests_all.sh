#!/bin/bash
# the whole GPU suite + smoke; output under gpurun_out/<tag>/
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-all}; rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --tb=short ${2:-} > $O/pytest_all.log 2>&1; echo "gpu tests rc=$?" | tee $O/summary.txt
tail -40 $O/pytest_all.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt; tail -2 $O/smoke.log
