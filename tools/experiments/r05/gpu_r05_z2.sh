#!/bin/bash
# round 5, call Z2: SQ counters of k_plane_pair / k_plane_work (month-grouped linear adjust)
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r05z2; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
sed -n '/^cat > \/tmp\/month_lin.py/,/^PY$/p' $GRAFT_REPO_ROOT/tools/experiments/r05/gpu_r05_z.sh | sed '1d;$d' > /tmp/month_lin.py
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/sq -o s -- python /tmp/month_lin.py > $O/log.txt 2>&1
python - <<'PY'
import csv, glob, collections, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r05z2"
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(O + "/sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        a = agg[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
with open(O + "/summary.txt", "w") as out:
    for k, v in agg.items():
        if "plane" not in k: continue
        line = k + " launches %d: " % v["SQ_WAVES"][0] + "  ".join("%s=%.4g" % (c, s / n) for c, (n, s) in sorted(v.items()))
        print(line); out.write(line + "\n")
PY
find $O -name "*.csv" -size +5M -delete
