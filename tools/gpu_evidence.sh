#!/bin/bash
# One pass over everything profiles/ cites (run on the GPU box: gpurun -- 'bash tools/gpu_evidence.sh <tag>').
# Writes under gpurun_out/<tag>_*; kernel_traffic.json is stamped with the hash of the kernel sources it was measured on.
tag=${1:-r06_z}
part=${2:-all}      # all | a (tests, smoke, the default line, headline kernel stats + PMC) | b (256-pocket stats + PMC, training, joint, per-kernel profiles): two gpurun calls of < 20 min
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out
set -o pipefail
run() { echo "== $*" >&2; timeout -k 10 "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: stopping" >&2; exit 1; fi; return $rc; }
if [ $part != b ]; then
run 900 python -m pytest tests -m gpu -q > $o/${tag}_gpu_tests.log 2>&1; tail -3 $o/${tag}_gpu_tests.log
run 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $o/${tag}_smoke.txt
# rocprofv3 kernel stats of the headline command alone (no other record of the default line), then the PMC passes (separate runs; 200-step chains)
run 400 rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --north-star-batch 0 --no-extra-shapes > /dev/null 2>&1
cp $(find $o/${tag}_stats -name "*kernel_stats.csv" | head -1) $o/${tag}_kernel_stats_b64_T1000.csv; rm -rf $o/${tag}_stats
BARGS="--steps 1 --warmup 0 --timesteps 200 --no-cpu-baseline --north-star-batch 0 --no-extra-shapes"
run 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $o/pmc_fetch -- python3 bench.py $BARGS > /dev/null 2> $o/pmc_fetch.err
run 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $o/pmc_write -- python3 bench.py $BARGS > /dev/null 2> $o/pmc_write.err
run 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS --output-format csv -d $o/pmc_sq -- python3 bench.py $BARGS > /dev/null 2> $o/pmc_sq.err
run 100 python3 tools/collect_traffic.py $o/pmc_fetch $o/pmc_write --sq $o/pmc_sq --out $o/kernel_traffic.json --command "python3 bench.py $BARGS" > $o/${tag}_pmc_summary.json
rm -rf $o/pmc_fetch $o/pmc_write $o/pmc_sq
# the traffic file belongs to these sources (stamped with their hash): the default line below reads it from profiles/
cp $o/kernel_traffic.json profiles/kernel_traffic.json
# the default line: headline (64 pockets, phar points inside the pocket), north_star_trained, fullatom_trained, drifted chain, fp32 engine, training step, cpu_baseline
run 900 python bench.py --steps 5 --warmup 2 > $o/${tag}_bench_b64.json 2> $o/${tag}_bench_b64.err
fi
if [ $part = a ]; then echo done part a; exit 0; fi
# the same counters and kernel stats at the north-star batch (256 pockets: 128-row edge kernels, 64-row node kernel), 100-step chains
B256="--batch 256 --steps 1 --warmup 0 --timesteps 100 --no-cpu-baseline --north-star-batch 0 --no-extra-shapes"
run 400 rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_stats256 -- python3 bench.py --batch 256 --steps 1 --warmup 1 --timesteps 300 --no-cpu-baseline --north-star-batch 0 --no-extra-shapes > /dev/null 2>&1
cp $(find $o/${tag}_stats256 -name "*kernel_stats.csv" | head -1) $o/${tag}_kernel_stats_b256_T300.csv; rm -rf $o/${tag}_stats256
run 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $o/pmc_fetch -- python3 bench.py $B256 > /dev/null 2> $o/pmc_fetch.err
run 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $o/pmc_write -- python3 bench.py $B256 > /dev/null 2> $o/pmc_write.err
run 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS --output-format csv -d $o/pmc_sq -- python3 bench.py $B256 > /dev/null 2> $o/pmc_sq.err
run 100 python3 tools/collect_traffic.py $o/pmc_fetch $o/pmc_write --sq $o/pmc_sq --batch 256 --out $o/kernel_traffic_b256.json --command "python3 bench.py $B256" > $o/${tag}_pmc_summary_b256.json
rm -rf $o/pmc_fetch $o/pmc_write $o/pmc_sq
cp $o/kernel_traffic_b256.json profiles/kernel_traffic_b256.json
# training step: throughput, kernel stats
for a in "--batch 64" "--batch 256" "--batch 64 --gemm bf16" "--batch 64 --no-pipeline"; do run 200 python tools/bench_train.py --steps 30 --warmup 5 $a 2>/dev/null | tail -1; done > $o/${tag}_train.jsonl
# what each of the round's training-step changes is worth on this box (options: side stream, half-engine forward)
bash tools/ab_train_opts.sh "wgrad_stream=0,train_half=0" "wgrad_stream=0" "train_half=0" "-" > $o/${tag}_train_options_ab.txt 2>&1
run 300 rocprofv3 --kernel-trace --output-format csv -d $o/${tag}_ttl -- python3 tools/bench_train.py --steps 6 --warmup 3 > /dev/null 2>&1
python3 tools/train_overlap.py $o/${tag}_ttl --chain > $o/${tag}_train_chain.txt 2>&1; rm -rf $o/${tag}_ttl
run 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_tstats -- python3 tools/bench_train.py --steps 4 --warmup 2 > /dev/null 2>&1
cp $(find $o/${tag}_tstats -name "*kernel_stats.csv" | head -1) $o/${tag}_train_kernel_stats_b64.csv; rm -rf $o/${tag}_tstats
run 200 python tools/bench_joint.py --batch 64 --timesteps 1000 2>/dev/null | tail -1 > $o/${tag}_joint.json
for a in "64" "256" "64 full-atom" "256 full-atom"; do run 100 python tools/steady_profile.py $a 2>/dev/null | tail -1; done > $o/${tag}_trained_geometry_profile.jsonl
echo done
