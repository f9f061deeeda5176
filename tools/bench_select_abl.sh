#!/bin/bash
# Ablation timing of k_select_lean (diagnostics: XH_SELECT_ABL skips phases, results are wrong on purpose).
export XH_DIAGNOSTICS=1  # the library ignores its diagnostic switches without it
for a in ${ABLS:-0 8 12 14 15}; do
  echo -n "abl=$a "; XH_SELECT_ABL=$a python tools/bench_eqm.py ${1:-10950} ${2:-103680} 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tminor_ms', round(d['quantile_tminor_ms'],3))"
done
