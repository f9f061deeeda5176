#!/bin/bash
# Round-6 profile collection on the GPU box: kernel-trace stats of the driver's bench command, HBM PMC passes (365-step grid
# and the full configurations separately: per-kernel means must not mix grid sizes), FETCH_SIZE calibration on the register
# sort's load pattern (tools/regsort_ubench), SQ + GRBM passes for the VALU-issue view.  PMC passes carry no tracing flags.
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/prof_r06; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o stats -- python $B --no-cpu > $O/stats.log 2>&1
grep -h "^{\"metric\"" $O/stats.log | tail -1 > $O/bench_line_under_rocprof.json
SMALL="python $B --steps 5 --warmup 1 --no-cpu --no-full"
FULL="python $B --steps 2 --warmup 1 --no-cpu --no-long"
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/small/pmc_fetch -o f -- $SMALL > $O/small_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/small/pmc_write -o w -- $SMALL > $O/small_write.log 2>&1
if [ "${XH_PROF_SKIP_FULL:-0}" != "1" ]; then  # (the 30-year kernels did not change since the committed profiles/r06/pmc_hbm_traffic_30yr.json)
timeout 1200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/full/pmc_fetch -o f -- $FULL > $O/full_fetch.log 2>&1
timeout 1200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/full/pmc_write -o w -- $FULL > $O/full_write.log 2>&1
fi
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/calib/pmc_fetch -o f -- $GRAFT_REPO_ROOT/tools/regsort_ubench > $O/calib_regsort.log 2>&1
if [ "${XH_PROF_SQ:-0}" = "small" ]; then  # the VALU-issue view of the 365-step kernels only (k_qdm_regsort changed in round 6)
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/sq -o s -- $SMALL > $O/sq.log 2>&1
timeout 900 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/grbm -o g -- $SMALL > $O/grbm.log 2>&1
fi
if [ "${XH_PROF_SQ:-0}" = "1" ]; then  # the VALU-issue view (tools/summarize_sq.py); the sort kernels did not change since profiles/r04/valu_busy.json
timeout 1200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/sq -o s -- $FULL > $O/sq.log 2>&1
timeout 1200 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/grbm -o g -- $FULL > $O/grbm.log 2>&1
fi
# keep only the csv summaries (the merge back is capped at 64 MiB)
find $O -name "*.csv" -size +20M -delete
find $O -type f ! -name "*.csv" ! -name "*.log" ! -name "*.json" -delete
du -sh $O
