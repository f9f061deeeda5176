#!/bin/bash
# PMC issue / wait breakdown of the secondary kernels (tools/bench_kernels.py) — run on the GPU box via gpurun
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/pmck_r02; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $GRAFT_REPO_ROOT/tools/bench_kernels.py > $O/bench_kernels.txt 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $O/pmc -o p -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py > $O/pmc.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py > $O/stats.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmck_r02"
agg=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0.0]))
for f in glob.glob(O+"/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0].replace("void ","")
        a=agg[k][r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
with open(O+"/pmc_summary.txt","w") as out:
    out.write("kernel  waves  VALU/wave  SALU/wave  VMEM/wave  active%  wait_inst%  wait_any%\n")
    for k,d in sorted(agg.items()):
        g=lambda n: d[n][1]/max(d[n][0],1) if n in d else 0.0
        wc=g("SQ_WAVE_CYCLES") or 1.0; w=g("SQ_WAVES") or 1.0
        out.write(f"{k[:70]:70s} {w:8.0f} {g('SQ_INSTS_VALU')/w:9.0f} {g('SQ_INSTS_SALU')/w:8.0f} {g('SQ_INSTS_VMEM_RD')/w:7.0f} {100*g('SQ_ACTIVE_INST_ANY')/wc:6.1f} {100*g('SQ_WAIT_INST_ANY')/wc:6.1f} {100*g('SQ_WAIT_ANY')/wc:6.1f}\n")
print(open(O+"/pmc_summary.txt").read())
PY
