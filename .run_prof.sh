cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "=== default bench"; timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 600 gpurun_out/bench_default.err; cut -c1-400 gpurun_out/bench_default.json
echo "=== rocprof stats"; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/bench_prof_b.log 2>&1; f=$(find gpurun_out/prof_b -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-160; rm -f $(find gpurun_out/prof_b -name "*kernel_trace.csv")
echo "=== pmc fetch"; rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --timesteps 60 --no-cpu-baseline > gpurun_out/pmc_fetch.log 2>&1; ls gpurun_out/pmc_fetch/*/ | head
echo "=== pmc write"; rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 1 --warmup 0 --timesteps 60 --no-cpu-baseline > gpurun_out/pmc_write.log 2>&1; ls gpurun_out/pmc_write/*/ | head
python3 - <<'PY'
import csv, glob, collections
for name in ('fetch', 'write'):
    fs = glob.glob(f'gpurun_out/pmc_{name}/*/*counter_collection.csv')
    print(name, fs)
    if not fs: continue
    agg = collections.defaultdict(lambda: [0.0, 0])
    with open(fs[0]) as f:
        r = csv.DictReader(f)
        for row in r:
            k = row.get('Kernel_Name', '')[:40]
            v = float(row.get('Counter_Value', 0))
            agg[(k, row.get('Counter_Name'))][0] += v; agg[(k, row.get('Counter_Name'))][1] += 1
    for (k, c), (s, n) in sorted(agg.items(), key=lambda x: -x[1][0])[:10]:
        print(f'  {k:42s} {c:12s} launches {n:6d} mean {s/n:12.2f}')
PY
rm -f $(find gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*kernel_trace.csv")
