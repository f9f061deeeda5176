#!/bin/bash
# round 5, call H: QDM cubic edge cases again + the bench line with the new extras (qdm_c4, eqm_doy_linear, c5_slab)
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r05h; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest -m gpu -q --tb=short tests/test_gpu_api.py -k "qdm or eqm" > $O/pytest.log 2>&1; echo "rc=$?" | tee $O/summary.txt
tail -5 $O/pytest.log
timeout 1200 python bench.py --no-cpu > $O/bench.log 2>&1
tail -1 $O/bench.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'])
for k,v in d['extra'].items():
    if isinstance(v,dict): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a in ('ms','frac','train_ms','adjust_ms','uploads')})
"
