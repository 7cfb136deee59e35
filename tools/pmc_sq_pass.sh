cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_LDS"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d gpurun_out/pmc_sq_ca64 -- python3 bench.py --steps 1 --warmup 0 --timesteps 100 --no-cpu-baseline --north-star-batch 0 > /dev/null 2> gpurun_out/pmc_sq_ca64.err
timeout -k 10 400 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d gpurun_out/pmc_sq_fa64 -- python3 bench.py --steps 1 --warmup 0 --timesteps 20 --representation full-atom --no-cpu-baseline --north-star-batch 0 > /dev/null 2> gpurun_out/pmc_sq_fa64.err
python3 - <<'PY'
import csv,glob,collections
for d in ('gpurun_out/pmc_sq_ca64','gpurun_out/pmc_sq_fa64'):
    acc=collections.defaultdict(lambda: collections.defaultdict(lambda:[0.0,0]))
    for f in glob.glob(d+'/**/*counter_collection.csv',recursive=True):
        for r in csv.DictReader(open(f)):
            n=r['Kernel_Name'].split('(')[0]
            a=acc[n][r['Counter_Name']]; a[0]+=float(r['Counter_Value']); a[1]+=1
    print('==',d)
    for n,cs in acc.items():
        if not any(k in n for k in ('k_edge_msg','k_node','k_edge_coord','k_embed')): continue
        print(n, {c:round(v[0]/v[1]) for c,v in cs.items()}, 'dispatches', list(cs.values())[0][1])
PY
rm -rf gpurun_out/pmc_sq_ca64 gpurun_out/pmc_sq_fa64
