#!/bin/bash
# same-box sweep of training-step options: bash tools/ab_train_sweep.sh "<opts>" "<opts>" ...  ("-" = defaults); prints ms per step at 64 / 128 / 256 complexes and with bf16 operands
cd $GRAFT_REPO_ROOT
run() { CMDGEN_OPTIONS=$1 timeout -k 10 120 python tools/bench_train.py --steps 20 --warmup 5 $2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f' % (d['ms_per_step']))"; }
for rep in 1 2; do for o in "$@"; do oo=$o; [ "$o" = "-" ] && oo=""; echo -n "[$o] "; for b in 64 128 256; do echo -n "B=$b "; run "$oo" "--batch $b" | tr '\n' ' '; done; echo -n "bf16 "; run "$oo" "--gemm bf16" | tr '\n' ' '; echo; done; done
