#!/bin/bash
# the driver's bench command (with the CPU baseline legs) on the GPU box -> gpurun_out/bench_r05/bench_line.json
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/bench_r05; rm -rf $O; mkdir -p $O
timeout 1500 python bench.py > $O/bench.log 2>&1; echo "rc=$?"
grep -h "^{\"metric\"" $O/bench.log | tail -1 > $O/bench_line.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_r05/bench_line.json"))
print("ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"], "cpu", (d.get("cpu_baseline") or {}).get("value"))
for k, v in d["extra"].items():
    if isinstance(v, dict):
        print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a in ("ms", "frac", "train_ms", "adjust_ms")})
print(json.dumps(d["extra"]["adapter_e2e"]))
PY
