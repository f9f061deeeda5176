#!/bin/bash
# SQ issue / wait counters of the benchmark's own kernels (bench.py, 365-step grid)
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_bench_sq; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu --no-full"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/p1 -o p -- $CMD > $O/p1.log 2>&1
python3 - <<'PY' | tee $O/summary.txt
import csv, glob, collections, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc_bench_sq"
agg=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0.0]))
for f in glob.glob(O+"/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0].replace("void ","")
        a=agg[k][r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
print("kernel  waves  VALU/wave  SALU/wave  LDS/wave  valu_busy_of_simd%  wait_inst%  wait_any%  waves/simd")
for k,d in sorted(agg.items()):
    g=lambda n: d[n][1]/max(d[n][0],1) if n in d else 0.0
    wc=g("SQ_WAVE_CYCLES") or 1.0; w=g("SQ_WAVES") or 1.0
    print(f"{k[:60]:60s} {w:8.0f} {g('SQ_INSTS_VALU')/w:9.0f} {g('SQ_INSTS_SALU')/w:8.0f} {g('SQ_INSTS_LDS')/w:7.0f} {100*g('SQ_ACTIVE_INST_VALU')/wc:6.1f}(per wave) {100*g('SQ_WAIT_INST_ANY')/wc:6.1f} {100*g('SQ_WAIT_ANY')/wc:6.1f}")
PY
rm -rf $O/p1
