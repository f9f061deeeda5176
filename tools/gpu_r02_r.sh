#!/bin/bash
# PMC of k_select_lean alone (tools/bench_lean2.py): where do the wave cycles go?
O=$GRAFT_REPO_ROOT/gpurun_out/r02ag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc1 -o p -- python $GRAFT_REPO_ROOT/tools/bench_lean2.py > $O/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU --output-format csv -d $O/pmc2 -o p -- python $GRAFT_REPO_ROOT/tools/bench_lean2.py > $O/pmc2.log 2>&1
python3 - <<'PY' | tee $O/summary.txt
import csv, glob, collections, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r02ag"
for d in ("pmc1","pmc2"):
    agg=collections.defaultdict(lambda:[0,0.0])
    for f in glob.glob(O+f"/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "select_lean" not in r["Kernel_Name"]: continue
            a=agg[r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
    for k,(n,v) in sorted(agg.items()): print(d, k, "launches", n, "mean %.4g"%(v/max(n,1)))
PY
rm -rf $O/pmc1 $O/pmc2
