#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + HBM PMC counters (separate passes, as the guide prescribes).
# Usage: tools/profile_bench.sh <tag>
set -u
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# kernel-trace statistics of the SAME command the driver runs (all extras incl. the 3650-step / 30-year configurations);
# the PMC passes use the 365-step grid only (per-kernel byte means would mix grid sizes otherwise)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python $GRAFT_REPO_ROOT/bench.py --no-cpu > $OUT/stats.log 2>&1
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu --no-full"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- $CMD > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -20
F=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
echo "== kernel stats ($F)"; head -25 "$F"
for f in $(find $OUT/pmc_fetch $OUT/pmc_write -name "*counter_collection.csv"); do
  echo "== $f"; python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    k = (r.get("Kernel_Name", "?")[:60], r.get("Counter_Name", "?"))
    agg[k][0] += 1
    agg[k][1] += float(r.get("Counter_Value", 0))
for (k, c), (n, v) in sorted(agg.items()):
    print(f"{k:60s} {c:12s} launches={n:4d} mean={v / n:.6g}")
PY
done
