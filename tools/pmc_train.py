#!/usr/bin/env python3
"""Per-kernel PMC summary of the TRAINING step (tools/bench_train.py) from three rocprofv3 passes (run on the GPU box):

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/tp_fetch -- python3 tools/bench_train.py --steps 2 --warmup 1 --no-pipeline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/tp_write -- python3 tools/bench_train.py --steps 2 --warmup 1 --no-pipeline
    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/tp_sq -- python3 tools/bench_train.py --steps 2 --warmup 1 --no-pipeline
    python3 tools/pmc_train.py gpurun_out/tp_fetch gpurun_out/tp_write gpurun_out/tp_sq > profiles/<name>.json

HBM-side bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of
16 B/lane reads, WRITE_SIZE is exact, both KiB; Infinity-Cache hits are counted: an upper bound on true HBM traffic)."""
import csv, glob, json, os, sys
from collections import defaultdict


def read(directory):
    out = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(directory, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f, newline='')):
            name = (row.get('Kernel_Name') or '').split('(')[0]
            if not name.replace('void ', '').startswith('k_'):
                continue
            slot = out[row.get('Counter_Name')][name.replace('void ', '')]
            slot[0] += float(row.get('Counter_Value') or 0.0); slot[1] += 1
    return out


fetch, write, sq = read(sys.argv[1]), read(sys.argv[2]), read(sys.argv[3])
res = {}
for k, (v, n) in fetch['FETCH_SIZE'].items():
    w = write['WRITE_SIZE'].get(k, [0.0, 1])
    res[k] = {'launches': n, 'read_MB_per_launch': round(2 * v * 1024 / n / 1e6, 2), 'written_MB_per_launch': round(w[0] * 1024 / max(w[1], 1) / 1e6, 2)}
for k, r in res.items():
    wc = sq['SQ_WAVE_CYCLES'].get(k)
    if wc and wc[0] > 0:
        r['wait_fraction'] = round(sq['SQ_WAIT_ANY'][k][0] / wc[0], 3)
        r['mfma_insts_per_launch'] = round(sq['SQ_INSTS_MFMA'][k][0] / wc[1])
        r['valu_insts_per_launch'] = round(sq['SQ_INSTS_VALU'][k][0] / wc[1])
order = sorted(res, key=lambda k: -(res[k]['read_MB_per_launch'] + res[k]['written_MB_per_launch']) * res[k]['launches'])
print(json.dumps({'command': 'python3 tools/bench_train.py --steps 2 --warmup 1 --no-pipeline (B=64 CA)', 'note': __doc__.split('\n\n')[-1].replace('\n', ' '),
                  'per_kernel': {k: res[k] for k in order}}, indent=1))
