cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "=== default bench"; timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; cut -c1-300 gpurun_out/bench_default.json
echo "=== B256 bench"; timeout 900 python bench.py --batch 256 --steps 2 --no-cpu-baseline > gpurun_out/bench_b256.json 2>/dev/null; cut -c1-200 gpurun_out/bench_b256.json
echo "=== FA bench"; timeout 900 python bench.py --batch 64 --representation full-atom --steps 1 --timesteps 200 --no-cpu-baseline > gpurun_out/bench_fa64.json 2>/dev/null; cut -c1-200 gpurun_out/bench_fa64.json
echo "=== rocprof stats"; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_d -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/bench_prof_d.log 2>&1; f=$(find gpurun_out/prof_d -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-130; rm -f $(find gpurun_out/prof_d -name "*kernel_trace.csv")
echo "=== pmc fetch"; rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch2 -- python3 bench.py --steps 1 --warmup 0 --timesteps 60 --no-cpu-baseline > gpurun_out/pmc_fetch2.log 2>&1
echo "=== pmc write"; rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write2 -- python3 bench.py --steps 1 --warmup 0 --timesteps 60 --no-cpu-baseline > gpurun_out/pmc_write2.log 2>&1
rm -f $(find gpurun_out/pmc_fetch2 gpurun_out/pmc_write2 -name "*kernel_trace.csv")
