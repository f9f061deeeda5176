#!/bin/bash
# round 4: FETCH_SIZE of the multi-year gather pattern — calibration on tools/gather_ubench (known bytes), then k_pdoy_quad
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r04p7; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/ub -o u -- $GRAFT_REPO_ROOT/tools/gather_ubench > $O/ub.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/quad -o q -- python $GRAFT_REPO_ROOT/tools/bench_pdoy30.py > $O/quad.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/quadw -o q -- python $GRAFT_REPO_ROOT/tools/bench_pdoy30.py > $O/quadw.log 2>&1
python3 - <<PY
import csv,glob,collections
for sub in ("ub","quad","quadw"):
    acc=collections.defaultdict(list)
    for f in glob.glob("$O/"+sub+"/**/*counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"][:70],r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        print(sub,k,len(v),"mean KiB",sum(v)/len(v),"-> GB x1",sum(v)/len(v)*1024/1e9)
PY
