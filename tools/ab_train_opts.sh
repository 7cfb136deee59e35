#!/bin/bash
# same-box A/B of training-step options: bash tools/ab_train_opts.sh "opts_a" "opts_b" ... (each a CMDGEN_OPTIONS string; "-" = defaults)
cd $GRAFT_REPO_ROOT
run() { CMDGEN_OPTIONS=$1 timeout -k 10 120 python tools/bench_train.py --steps 30 --warmup 5 $2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms  loss %.7f' % (d['ms_per_step'], d['last_loss']))"; }
for rep in 1 2; do for o in "$@"; do oo=$o; [ "$o" = "-" ] && oo=""; echo -n "[B=64 f32 $o] "; run "$oo" ""; done; done
for o in "$@"; do oo=$o; [ "$o" = "-" ] && oo=""; echo -n "[B=64 bf16 $o] "; run "$oo" "--gemm bf16"; done
for o in "$@"; do oo=$o; [ "$o" = "-" ] && oo=""; echo -n "[B=256 f32 $o] "; run "$oo" "--batch 256"; done
