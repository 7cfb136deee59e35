#!/bin/bash
# SQ counters of the full-atom edge kernel (64 full-atom pockets, 20-step chain of the bounded-schedule model): tools/pmc_fullatom.sh <tag>
tag=${1:-r04_w}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out
FA="--batch 64 --representation full-atom --steps 1 --warmup 0 --timesteps 20 --no-cpu-baseline --north-star-batch 0 --no-extra-shapes"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_fa_stats -- python3 bench.py $FA > /dev/null 2>&1
cp $(find $o/${tag}_fa_stats -name "*kernel_stats.csv" | head -1) $o/${tag}_kernel_stats_fullatom_b64_T20.csv; rm -rf $o/${tag}_fa_stats
timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS --output-format csv -d $o/pmc_sq -- python3 bench.py $FA > /dev/null 2> $o/pmc_sq.err
python3 - <<'PY' > $o/${tag}_sq_fullatom_b64.json
import csv, glob, json, collections, os
o = os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'gpurun_out')
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(o, 'pmc_sq', '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f, newline='')):
        n = r.get('Kernel_Name') or ''
        key = 'k_edge128<msg>' if 'k_edge128<false>' in n else 'k_edge128<coord>' if 'k_edge128<true>' in n else 'k_node64' if 'k_node64' in n else None
        if key:
            s = acc[key][r['Counter_Name']]; s[0] += float(r['Counter_Value']); s[1] += 1
print(json.dumps({k: {c: v[0] / max(v[1], 1) for c, v in d.items()} for k, d in acc.items()}, indent=1))
PY
rm -rf $o/pmc_sq
