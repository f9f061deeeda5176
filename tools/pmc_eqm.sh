#!/bin/bash
# PMC counters for the select kernels (EQM train).  Usage: tools/pmc_eqm.sh <tag> [G]
export XH_DIAGNOSTICS=1  # the library ignores its diagnostic switches without it
TAG=${1:-x}; export XH_SELECT_G=${2:-32}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/tools/bench_eqm.py ${3:-365} ${4:-1036800}"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/a -o a -- $CMD > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/b -o b -- $CMD > $OUT/b.log 2>&1
for f in $(find $OUT -name "*counter_collection.csv"); do
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    if "select" not in r.get("Kernel_Name", ""): continue
    k = (r["Kernel_Name"][:44], r["Counter_Name"])
    agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
for (k, c), (n, v) in sorted(agg.items()):
    print(f"{k:44s} {c:22s} n={n:3d} mean={v / n:.6g}")
PY
done
tail -3 $OUT/a.log | cut -c1-200
