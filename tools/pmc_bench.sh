#!/bin/bash
# SQ instruction counters for the bench kernels (own PMC passes, no tracing).  Usage: tools/pmc_bench.sh <tag>
TAG=${1:-x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcb_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/a -o a -- $CMD > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD --output-format csv -d $OUT/b -o b -- $CMD > $OUT/b.log 2>&1
for f in $(find $OUT -name "*counter_collection.csv"); do
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    k = (r.get("Kernel_Name", "")[:40], r["Counter_Name"])
    agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
for (k, c), (n, v) in sorted(agg.items()):
    if "rocclr" in k or "fill" in k or "correction" in k or "missing" in k: continue
    print(f"{k:40s} {c:24s} n={n:3d} mean={v / n:.6g}")
PY
done
