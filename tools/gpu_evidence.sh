#!/bin/bash
# One pass over everything profiles/ cites (run on the GPU box: gpurun -- 'bash tools/gpu_evidence.sh <tag>').
# Writes under gpurun_out/<tag>_*; kernel_traffic.json is stamped with the hash of the kernel sources it was measured on.
tag=${1:-r03_z}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out
set -o pipefail
run() { echo "== $*" >&2; timeout -k 10 "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: stopping" >&2; exit 1; fi; return $rc; }
run 700 python -m pytest tests -m gpu -q > $o/${tag}_gpu_tests.log 2>&1; tail -3 $o/${tag}_gpu_tests.log
run 300 python -m pytest tests/test_hip_split.py -q -s 2>&1 | grep -E "max\|d eps\||RMS|passed|failed" > $o/${tag}_split_tests.log
run 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $o/${tag}_smoke.txt
run 300 python bench.py --steps 5 --warmup 2 > $o/${tag}_bench_b64.json 2> $o/${tag}_bench_b64.err
BARGS="--steps 1 --warmup 0 --timesteps 200 --no-cpu-baseline --north-star-batch 0 --no-extra-shapes"     # PMC passes serialise every dispatch: a 200-step chain is plenty
run 400 rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --north-star-batch 0 --no-extra-shapes > /dev/null 2>&1
cp $(find $o/${tag}_stats -name "*kernel_stats.csv" | head -1) $o/${tag}_kernel_stats_b64_T1000.csv; rm -rf $o/${tag}_stats
run 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $o/pmc_fetch -- python3 bench.py $BARGS > /dev/null 2> $o/pmc_fetch.err
run 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $o/pmc_write -- python3 bench.py $BARGS > /dev/null 2> $o/pmc_write.err
run 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_LDS --output-format csv -d $o/pmc_sq -- python3 bench.py $BARGS > /dev/null 2> $o/pmc_sq.err
run 100 python3 tools/collect_traffic.py $o/pmc_fetch $o/pmc_write --sq $o/pmc_sq --out $o/kernel_traffic.json --command "python3 bench.py $BARGS" > $o/${tag}_pmc_summary.json
rm -rf $o/pmc_fetch $o/pmc_write $o/pmc_sq
run 300 python bench.py --batch 256 --steps 2 --warmup 1 --north-star-batch 0 --no-cpu-baseline > $o/${tag}_bench_b256.json 2>/dev/null
run 300 python bench.py --representation full-atom --batch 64 --timesteps 200 --steps 1 --warmup 1 --north-star-batch 0 --cpu-seconds 8 > $o/${tag}_bench_fullatom_b64.json 2>/dev/null
run 400 python bench.py --representation full-atom --batch 256 --timesteps 100 --steps 1 --warmup 1 --north-star-batch 0 --no-cpu-baseline > $o/${tag}_bench_fullatom_b256.json 2>/dev/null
# the same three workloads on the fp32 matrix instruction (cmdgen_set_gemm_mode(0)) next to the default split-bf16 engine
run 300 python bench.py --gemm fp32 --batch 256 --steps 2 --warmup 1 --north-star-batch 0 --no-cpu-baseline > $o/${tag}_bench_b256_fp32engine.json 2>/dev/null
run 300 python bench.py --gemm fp32 --representation full-atom --batch 64 --timesteps 200 --steps 1 --warmup 1 --north-star-batch 0 --no-cpu-baseline > $o/${tag}_bench_fullatom_b64_fp32engine.json 2>/dev/null
run 300 python bench.py --gemm fp32 --steps 3 --warmup 1 --north-star-batch 0 --no-cpu-baseline > $o/${tag}_bench_b64_fp32engine.json 2>/dev/null
for a in "--batch 64" "--batch 256" "--batch 256 --gemm bf16" "--batch 64 --gemm bf16" "--batch 64 --no-pipeline" "--batch 256 --no-pipeline"; do run 200 python tools/bench_train.py --steps 30 --warmup 5 $a 2>/dev/null | tail -1; done > $o/${tag}_train.jsonl
run 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_tstats -- python3 tools/bench_train.py --steps 4 --warmup 2 > /dev/null 2>&1
cp $(find $o/${tag}_tstats -name "*kernel_stats.csv" | head -1) $o/${tag}_train_kernel_stats_b64.csv; rm -rf $o/${tag}_tstats
# PMC passes over the training step (HBM-side bytes per launch, wait fractions) and a per-launch timeline of one step
TA="--steps 2 --warmup 1 --no-pipeline"
run 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $o/tp_fetch -- python3 tools/bench_train.py $TA > /dev/null 2>&1
run 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $o/tp_write -- python3 tools/bench_train.py $TA > /dev/null 2>&1
run 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $o/tp_sq -- python3 tools/bench_train.py $TA > /dev/null 2>&1
run 100 python3 tools/pmc_train.py $o/tp_fetch $o/tp_write $o/tp_sq > $o/${tag}_train_pmc.json; rm -rf $o/tp_fetch $o/tp_write $o/tp_sq
run 300 rocprofv3 --kernel-trace --output-format csv -d $o/tl -- python3 tools/bench_train.py --steps 4 --warmup 2 --no-pipeline > /dev/null 2>&1
run 100 python3 tools/train_timeline.py $o/tl > $o/${tag}_train_timeline.txt; rm -rf $o/tl
run 200 python tools/bench_joint.py --batch 64 --timesteps 1000 2>/dev/null | tail -1 > $o/${tag}_joint.json
for b in 64 256; do run 100 python tools/steady_profile.py $b 2>/dev/null | tail -1; done > $o/${tag}_trained_geometry_profile.jsonl
run 100 python tools/steady_profile.py 64 full-atom 2>/dev/null | tail -1 >> $o/${tag}_trained_geometry_profile.jsonl
echo done
