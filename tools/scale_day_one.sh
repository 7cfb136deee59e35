#!/bin/bash
# First contact with an 8-GPU node (BASELINE configs[2] / [3]): the whole scaling protocol in one go, no edits needed.
#   tools/scale_day_one.sh [--dry-run-launch]     (--dry-run-launch: launch logic only, on CPU over gloo - what tests/test_bench_launch.py runs)
# Every line it prints carries n_gpus as COUNTED by an all-reduce of ones over the process group and, on GPUs, "collectives": "nccl ..."
# (RCCL).  Ranks are started by bench.py / bench_train.py themselves from a process that has not touched the GPU (torch.distributed.run
# children; never a re-exec), rendezvous on 127.0.0.1.
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
extra="${1:-}"
out=${SCALE_OUT:-gpurun_out/scale_day_one}
mkdir -p "$out"
rc=0
for n in ${SCALE_NS:-1 2 4 8}; do
  echo "== weak scaling (64 pockets per GPU, configs[1] per rank), N=$n"
  python bench.py --gpus $n --steps 3 --warmup 1 --no-extra-shapes --no-cpu-baseline $extra | tee "$out/weak_n$n.json" || rc=1
done
for n in ${SCALE_NS:-1 2 4 8}; do
  echo "== strong scaling (configs[2]: ONE batch of 512 pockets over N GPUs), N=$n"
  python bench.py --gpus $n --strong --global-batch 512 --steps 3 --warmup 1 --no-extra-shapes --no-cpu-baseline $extra | tee "$out/strong_n$n.json" || rc=1
done
if [ "$extra" != "--dry-run-launch" ]; then
  echo "== training (configs[3]): data-parallel step on 8 GPUs, bf16 GEMM operands, chunked all-reduce overlapped with the backward pass"
  for g in bf16 fp32; do
    port=$((20000 + RANDOM % 20000))
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port $port tools/bench_train.py --gpus 8 --gemm $g | tee "$out/train_n8_$g.json" || rc=1
  done
fi
exit $rc
