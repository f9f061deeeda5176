#!/bin/bash
# round 4, percentile_doy on 30 years: pure kernel durations (rocprofv3 kernel trace) + SQ counters, quad vs top16
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r04p3; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export XH_DIAGNOSTICS=1
B="python $GRAFT_REPO_ROOT/tools/bench_pdoy30.py"
for v in quad top16; do
  q=1; [ $v = top16 ] && q=0
  XH_PDOY_QUAD=$q timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v/stats -o s -- $B > $O/$v.log 2>&1
  XH_PDOY_QUAD=$q timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/$v/sq -o s -- $B > $O/${v}_sq.log 2>&1
  XH_PDOY_QUAD=$q timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_F64 --output-format csv -d $O/$v/sq2 -o s -- $B > $O/${v}_sq2.log 2>&1
done
find $O -name "*.csv" -size +20M -delete
find $O -type f ! -name "*.csv" ! -name "*.log" ! -name "*.json" -delete
for v in quad top16; do echo == $v; grep -h "pdoy" $(find $O/$v/stats -name "*kernel_stats.csv") | head -3; done
python3 - <<'PY'
import csv,glob,os,collections
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04p3'
for v in ('quad','top16'):
    for sub in ('sq','sq2'):
        for f in glob.glob(f'{O}/{v}/{sub}/**/*counter_collection.csv',recursive=True):
            acc=collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if 'pdoy' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
            print(v,sub,{k:sum(x)/len(x) for k,x in acc.items()})
PY
