#!/usr/bin/env python3
"""Turn rocprofv3 PMC passes of bench.py into profiles/kernel_traffic.json (-> bench.py's `roofline.traffic`).

The passes (separate runs, TCC slots do not hold both counters; program directly after `--`):

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --north-star-batch 0
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --north-star-batch 0
    python3 tools/collect_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write --out gpurun_out/kernel_traffic.json [--sq gpurun_out/pmc_sq]

HBM-side bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE reports
half the bytes of 16 B/lane reads, WRITE_SIZE is exact; both in KiB; Infinity-Cache hits are counted, so this is an upper
bound on true HBM traffic).  The file is stamped with the hash of the kernel sources it was measured on (bench.py refuses
a stale one) and with the batch / representation of the workload.
"""
import argparse
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# kernel key -> substrings of the kernel name (the 128-row edge kernels of kernels_edge128.hip are k_edge128<false> = messages, <true> = coordinates)
KERNELS = {'edge_msg': ('k_edge_msg', 'k_edge128<false>'), 'node': ('k_node',), 'edge_coord': ('k_edge_coord', 'k_edge128<true>'), 'embed': ('k_embed',),
           'readout': ('k_readout',), 'ddpm_step': ('k_ddpm_step',), 'edge_count': ('k_edge_count',), 'edge_write': ('k_edge_write',),
           'write_embed': ('k_write_embed',), 'step_count': ('k_step_count',)}


def read_counters(directory):
    """-> {counter: {kernel key: (sum, dispatches)}} from every *counter_collection.csv below `directory`."""
    out = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    files = glob.glob(os.path.join(directory, '**', '*counter_collection.csv'), recursive=True)
    if not files:
        raise SystemExit(f'no *counter_collection.csv under {directory}')
    for f in files:
        with open(f, newline='') as fh:
            for row in csv.DictReader(fh):
                name = row.get('Kernel_Name') or row.get('Kernel-Name') or ''
                key = next((k for k, pats in KERNELS.items() if any(pat in name for pat in pats)), None)
                if key is None:
                    continue
                c = row.get('Counter_Name') or row.get('Counter-Name')
                v = float(row.get('Counter_Value') or row.get('Counter-Value') or 0.0)
                slot = out[c][key]
                slot[0] += v
                slot[1] += 1
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('fetch_dir')
    ap.add_argument('write_dir')
    ap.add_argument('--sq', default=None, help='optional pass with SQ counters (SQ_WAIT_ANY, SQ_WAVE_CYCLES, SQ_INSTS_MFMA, ...)')
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'kernel_traffic.json'))
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--representation', default='CA')
    ap.add_argument('--command', default='python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --north-star-batch 0')
    a = ap.parse_args()
    from bench import kernel_source_sha
    fetch = read_counters(a.fetch_dir).get('FETCH_SIZE', {})
    write = read_counters(a.write_dir).get('WRITE_SIZE', {})
    per, detail = {}, {}
    for k in KERNELS:
        if k in fetch or k in write:
            f = fetch.get(k, [0.0, 1]); w = write.get(k, [0.0, 1])
            fk, wk = f[0] / max(f[1], 1), w[0] / max(w[1], 1)
            per[k] = (2.0 * fk + wk) * 1024.0
            detail[k] = {'mean_FETCH_SIZE_KiB': fk, 'mean_WRITE_SIZE_KiB': wk, 'dispatches': int(max(f[1], w[1]))}
    doc = {'workload_batch': a.batch, 'representation': a.representation, 'kernel_source_sha': kernel_source_sha(),
           'hbm_bytes_per_launch': per, 'per_kernel_counters': detail,
           'source': f'rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- {a.command}',
           'formula': '(2*FETCH_SIZE + WRITE_SIZE)*1024 per dispatch: gfx950 FETCH_SIZE reports half of 16 B/lane reads; '
                      'includes Infinity-Cache hits (upper bound on HBM bytes)'}
    if a.sq:
        sq = read_counters(a.sq)
        doc['sq_counters_mean_per_dispatch'] = {c: {k: v[0] / max(v[1], 1) for k, v in d.items()} for c, d in sq.items()}
        wa, wc = sq.get('SQ_WAIT_ANY', {}), sq.get('SQ_WAVE_CYCLES', {})
        doc['wait_fraction'] = {k: wa[k][0] / wc[k][0] for k in wa if k in wc and wc[k][0] > 0}
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(doc, open(a.out, 'w'), indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == '__main__':
    main()
