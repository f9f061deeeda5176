#!/bin/bash
# kernel timeline of the config-4 train pipeline (do the transposes and the selection overlap?)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02u; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $GRAFT_REPO_ROOT/tools/bench_c4.py 243200 > $O/run.log 2>&1
tail -2 $O/run.log
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' | tee $O/timeline.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    n = r["Kernel_Name"]
    if "transpose" in n or "select_lean" in n or "eqm" in n.lower():
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "T" if "transpose" in n else ("S" if "select_lean" in n else "O"), r.get("Stream_Id", "")))
ev.sort()
t0 = ev[0][0]
for s, e, k, st in ev[-60:]:
    print(f"{k} start {(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f} us  stream {st}")
PY
rm -rf $O/trace
