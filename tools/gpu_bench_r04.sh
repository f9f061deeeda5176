#!/bin/bash
# the driver's bench command (+ the full-size extras) on the GPU box
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-bench_r04}; rm -rf $O; mkdir -p $O
timeout 1500 python bench.py ${2:-} > $O/bench.out 2> $O/bench.err; echo "rc=$?" | tee $O/summary.txt
tail -c 300 $O/bench.err
python - <<PY
import json
r = json.loads([l for l in open("$O/bench.out") if l.startswith("{")][-1])
print("value", r["value"], "ms/step", r["ms_per_step"], "frac", r["roofline"]["frac"])
for k, v in r.get("extra", {}).items():
    if isinstance(v, dict):
        print(f"  {k:24s}", {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a in ("ms", "frac", "train_ms", "adjust_ms", "train_frac", "percentile_doy_ms", "threshold_count_ms")})
print(json.dumps(r["extra"].get("adapter_e2e"), indent=1)[:1500])
PY
