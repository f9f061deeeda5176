#!/bin/bash
# bench.py exactly as the driver runs it (N = 1); the line lands in gpurun_out/<tag>/bench_line.json
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-bench}; mkdir -p $O
timeout 1200 python bench.py > $O/bench_line.json 2> $O/bench_err.log; echo "bench rc=$?"
python3 - $O/bench_line.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.4g %s  ms/step %.3f  frac %.3f  traffic_source %s" % (r["value"], r["unit"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"].get("traffic_source")))
print("cpu", r["cpu_baseline"])
for k, v in r["extra"].items():
    print("%-24s ms %8.3f  frac %.3f" % (k, v["ms"], v.get("frac", float("nan"))), {kk: round(vv, 3) for kk, vv in v.items() if kk.endswith("_ms")})
PY
