#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02c; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --tb=short -x -k "register_sort or one_year or quantile_series or eqm" > $O/pytest_new.log 2>&1; echo "new tests rc=$?" | tee -a $O/summary.txt
tail -5 $O/pytest_new.log
for a in 0 1 2 3 4 7; do
  echo "abl=$a"; XH_DIAGNOSTICS=1 XH_REGSORT_ABL=$a timeout 300 python tools/bench_eqm.py 365 2>&1 | tail -1 | cut -c1-120
done | tee $O/abl.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc1 -o p -- python $GRAFT_REPO_ROOT/tools/bench_eqm.py 365 > $O/pmc1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_IFETCH --output-format csv -d $O/pmc2 -o p -- python $GRAFT_REPO_ROOT/tools/bench_eqm.py 365 > $O/pmc2.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_IFETCH_LEVEL SQ_INSTS_VALU --output-format csv -d $O/pmc3 -o p -- python $GRAFT_REPO_ROOT/tools/bench_eqm.py 365 > $O/pmc3.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r02c"
for d in ("pmc1","pmc2","pmc3"):
    for f in glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True):
        agg=collections.defaultdict(lambda:[0,0.0])
        for r in csv.DictReader(open(f)):
            k=r.get("Kernel_Name","?")
            if "regsort" not in k: continue
            c=r.get("Counter_Name"); agg[c][0]+=1; agg[c][1]+=float(r.get("Counter_Value",0))
        for c,(n,v) in sorted(agg.items()): print(d, c, "launches",n,"mean",v/n)
PY
