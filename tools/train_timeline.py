#!/usr/bin/env python3
"""Where one training step's wall time goes, from a rocprofv3 kernel trace of tools/bench_train.py:

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 tools/bench_train.py --steps 4 --warmup 2
    python3 tools/train_timeline.py gpurun_out/tl

Steps are delimited by k_adamw.  For the last full step: busy time, idle time, and the idle gaps grouped by the kernel
that FOLLOWS them (the kernel the GPU was waiting for the host to launch)."""
import csv, glob, sys, collections

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
ends = [i for i, r in enumerate(rows) if r[2].startswith('k_adamw')]
lo, hi = ends[-2] + 1, ends[-1] + 1
step = rows[lo:hi]
t0, t1 = rows[ends[-2]][1], step[-1][1]
busy = sum(e - s for s, e, _ in step)
print(f'step wall {1e-3 * (t1 - t0):.0f} us, busy {1e-3 * busy:.0f} us, launches {len(step)}')
gaps = collections.defaultdict(lambda: [0, 0])
prev = t0
phase = collections.OrderedDict()
for s, e, n in step:
    g = max(0, s - prev)
    key = n.split('(')[0][:70]
    gaps[key][0] += g; gaps[key][1] += 1
    prev = max(prev, e)
for k, (g, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f'{1e-3 * g:8.1f} us idle before {c:4d} x {k}')
# coarse phases: up to the first k_embed / k_write_embed (prologue), to k_readout (forward), to first train-backward kernel, rest
names = [n for _, _, n in step]
def first(pred, start=0):
    for i in range(start, len(step)):
        if pred(names[i]): return i
    return len(step) - 1
i_fwd = first(lambda n: 'k_edge_count' in n or 'k_edge_write' in n)
i_ro = first(lambda n: 'k_readout' in n, i_fwd)
i_bwd = first(lambda n: 'k_eps_bwd' in n or 'k_train_loss' in n, i_ro)
marks = [('prologue (host loss setup, noising)', 0, i_fwd), ('forward', i_fwd, i_ro + 1), ('loss assembly', i_ro + 1, i_bwd), ('backward + optimizer', i_bwd, len(step))]
for name, a, b in marks:
    if b <= a: continue
    ws = (step[a - 1][1] if a else t0); we = step[b - 1][1]
    bz = sum(e - s for s, e, _ in step[a:b])
    print(f'{name:40s} wall {1e-3 * (we - ws):8.0f} us  busy {1e-3 * bz:8.0f} us  launches {b - a}')
# every launch of the kernels named on the command line (after the trace directory), in order: duration and grid
if len(sys.argv) > 2:
    full = list(csv.DictReader(open(f)))
    full.sort(key=lambda r: int(r['Start_Timestamp']))
    full = full[lo:hi]
    for pat in sys.argv[2:]:
        print('--', pat)
        for r in full:
            if pat in r['Kernel_Name']:
                print(f"{1e-3 * (int(r['End_Timestamp']) - int(r['Start_Timestamp'])):8.1f} us  grid {r.get('Grid_Size', '?'):>8s} wg {r.get('Workgroup_Size', '?'):>5s}  {r['Kernel_Name'][:60]}")
# per-kernel totals of the step
tot = collections.defaultdict(lambda: [0, 0])
for s_, e_, n_ in step:
    k = n_.split('(')[0][:60]
    tot[k][0] += e_ - s_; tot[k][1] += 1
print('-- per kernel, this step')
for k, (t, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:30]:
    print(f'{1e-3 * t:8.1f} us {c:4d} x {k}')
