#!/bin/bash
# round 6, call F: QDM at config 4 after the LDS staging of the epilogue's inputs — time, HBM traffic of k_hs_collect<8,5,true>
# (FETCH_SIZE / WRITE_SIZE in separate passes), bitwise fuzz
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r06_f; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 400 python tools/experiments/r06/qdm_c4_abl.py 0 128 > $O/qdm_abl.txt 2>&1; tail -1 $O/qdm_abl.txt
FUZZ_ONLY=qdm FUZZ_SEED=617 timeout 300 python tools/fuzz_r05.py 100 > $O/fuzz_qdm.txt 2>&1; tail -1 $O/fuzz_qdm.txt
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o p -- python $GRAFT_REPO_ROOT/tools/experiments/r06/qdm_c4_abl.py 0 > $O/pmc_$c.log 2>&1
done
python - <<PY
import csv, glob
for c, mul in (("FETCH_SIZE", 2), ("WRITE_SIZE", 1)):
    f = glob.glob("$O/pmc_%s/**/*counter_collection.csv" % c, recursive=True)
    agg = {}
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0][-60:]
        agg.setdefault(k, []).append(float(r["Counter_Value"]) * 1024 * mul / 1e9)
    for k, v in sorted(agg.items()):
        print(c, k, len(v), "max %.2f GB" % max(v))
PY
find $O -name "*.csv" -size +5M -delete
