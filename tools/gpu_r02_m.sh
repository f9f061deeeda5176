#!/bin/bash
cd $GRAFT_REPO_ROOT
export XH_DIAGNOSTICS=1
O=gpurun_out/r02v; mkdir -p $O
XH_SELECT_PROF=1 timeout 300 python tools/bench_c4.py 24320 > $O/glds.log 2>&1
XH_LEAN_NOGLDS=1 XH_SELECT_PROF=1 timeout 300 python tools/bench_c4.py 24320 > $O/noglds.log 2>&1
tail -13 $O/glds.log; tail -13 $O/noglds.log
