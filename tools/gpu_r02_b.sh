#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02b; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --tb=short -k "register_sort or one_year or multi_year or quantile_series or eqm or percentile_doy or doy or tx90p or 30yr or config5" > $O/pytest_new.log 2>&1; echo "new tests rc=$?" | tee -a $O/summary.txt
tail -15 $O/pytest_new.log
timeout 600 python tools/bench_eqm.py 365 > $O/eqm365_regsort.json 2> $O/eqm365_regsort.err; cat $O/eqm365_regsort.json
XH_DIAGNOSTICS=1 XH_REGSORT_IRREGULAR=1 timeout 600 python tools/bench_eqm.py 365 > $O/eqm365_irregular.json 2>&1; cat $O/eqm365_irregular.json
timeout 1500 python -m pytest tests -m gpu -q --tb=short > $O/pytest_all.log 2>&1; echo "all tests rc=$?" | tee -a $O/summary.txt
tail -8 $O/pytest_all.log
timeout 900 python bench.py --no-cpu > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02b/bench.json"))
for k,v in d["extra"].items(): print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ("ms","frac","train_ms","adjust_ms","percentile_doy_ms","threshold_count_ms")})
PY
XH_BENCH_FORCE_DIST=1 timeout 600 python bench.py --no-cpu --no-extra --steps 10 > $O/bench_dist1.json 2> $O/bench_dist1.err; echo "bench dist(1 rank, RCCL via C ABI) rc=$?" | tee -a $O/summary.txt; cut -c1-400 $O/bench_dist1.json; tail -3 $O/bench_dist1.err
XH_BENCH_FORCE_DIST=1 XH_BENCH_SYNC_GATHER=1 timeout 600 python bench.py --no-cpu --no-extra --steps 10 > $O/bench_dist1_sync.json 2> $O/bench_dist1_sync.err; echo "bench dist sync rc=$?" | tee -a $O/summary.txt; cut -c1-300 $O/bench_dist1_sync.json
XH_BENCH_FORCE_DIST=1 timeout 900 python bench.py --workload c5 --steps 2 --warmup 1 > $O/bench_c5.json 2> $O/bench_c5.err; echo "bench c5 rc=$?" | tee -a $O/summary.txt; cat $O/bench_c5.json; tail -3 $O/bench_c5.err
python -c "import sys; print('torch' in sys.modules)"
