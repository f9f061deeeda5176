#!/bin/bash
# round 5: the whole GPU suite on two other draws of the seeded inputs (XH_TEST_SEED)
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_seeds; rm -rf $O; mkdir -p $O
for s in 7 20260926; do
  XH_TEST_SEED=$s timeout 1200 python -m pytest tests -m gpu -q --tb=line > $O/pytest_$s.log 2>&1; echo "seed $s rc=$?" | tee -a $O/summary.txt
  grep -E "^FAILED|^/root.*Error|passed|failed" $O/pytest_$s.log | tail -15
done
