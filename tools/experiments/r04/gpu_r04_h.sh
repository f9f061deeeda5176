#!/bin/bash
# (record of an experiment: the XH_PDOY_NONP1 switch and the NP1 instance left the tree with it)
# round 4, headline kernel k_pdoy_slide<5,4>: the single-percentile instance (stores left in flight across the step's join)
# against the generic one (XH_PDOY_NONP1=1), alternating in one call; then the one-year percentile tests
set -u
cd $GRAFT_REPO_ROOT
export XH_DIAGNOSTICS=1
for rep in 1 2 3; do
  for v in 0 1; do
    if [ $v = 1 ]; then export XH_PDOY_NONP1=1; else unset XH_PDOY_NONP1; fi
    python bench.py --no-cpu --no-full --steps 50 --warmup 10 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('nonp1=$v', 'ms_per_step', round(d['ms_per_step'],4), 'pdoy_ms', round(d['roofline']['ms'],4), 'frac', round(d['roofline']['frac'],4))"
  done
done
unset XH_PDOY_NONP1
timeout 600 python -m pytest -m gpu -q --tb=short -x tests -k "doy or tx90 or percentile" 2>&1 | tail -2
