#!/bin/bash
# same-box A/B of two builds of the library on the training step: bash tools/ab_train_libs.sh <libA.so> <libB.so>   ("-" = the in-tree library)
cd $GRAFT_REPO_ROOT
run() { lib=$1; shift; if [ "$lib" = "-" ]; then unset CMDGEN_LIB; else export CMDGEN_LIB=$lib; fi
  timeout -k 10 120 python tools/bench_train.py --steps 40 --warmup 5 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms' % d['ms_per_step'])"; }
for rep in 1 2 3; do for l in "$@"; do echo -n "[B=64 f32 $l] "; run $l; done; done
for rep in 1 2; do for l in "$@"; do echo -n "[B=64 bf16 $l] "; run $l --gemm bf16; done; done
for l in "$@"; do echo -n "[B=256 f32 $l] "; run $l --batch 256; done
