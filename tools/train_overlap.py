#!/usr/bin/env python3
"""Overlap view of one training step from a rocprofv3 kernel trace (two streams: the chain of data gradients and the weight gradients
beside it): per kernel name, launches, summed duration, and how much of that duration some OTHER kernel was running too.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 tools/bench_train.py --steps 6 --warmup 3
    python3 tools/train_overlap.py gpurun_out/tl"""
import csv, glob, sys, collections

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:48], r.get('Queue_Id', '?')) for r in csv.DictReader(open(f))]
rows.sort()
ends = [i for i, r in enumerate(rows) if r[2].startswith('k_adamw')]
lo, hi = ends[-2] + 1, ends[-1] + 1
step = rows[lo:hi]
t0, t1 = rows[ends[-2]][1], step[-1][1]
# union of busy intervals
iv = sorted((s, e) for s, e, _, _ in step)
busy = 0; cur_s, cur_e = iv[0]
for s, e in iv[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f'step wall {1e-3 * (t1 - t0):.0f} us, any-kernel-busy {1e-3 * busy:.0f} us, sum of durations {1e-3 * sum(e - s for s, e, _, _ in step):.0f} us, launches {len(step)}, queues {sorted(set(q for *_, q in step))}')
tot = collections.defaultdict(lambda: [0, 0, 0])
for i, (s, e, n, q) in enumerate(step):
    ov = 0
    for j, (s2, e2, n2, q2) in enumerate(step):
        if i != j and s2 < e and e2 > s: ov += min(e, e2) - max(s, s2)
    tot[(n, q)][0] += 1; tot[(n, q)][1] += e - s; tot[(n, q)][2] += min(ov, e - s)
for (n, q), (c, d, ov) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f'{1e-3 * d:8.1f} us {c:4d} x  overlapped {1e-3 * ov:7.1f} us  q{q}  {n}')

# the main queue's chain in launch order (option --chain): start offset, duration, gap before it, kernel; side-queue launches marked '|'
if '--chain' in sys.argv:
    mainq = max(set(q for *_, q in step), key=lambda q: sum(1 for r in step if r[3] == q))
    prev = t0
    for s_, e_, n_, q_ in step:
        if q_ == mainq:
            print(f'{1e-3 * (s_ - t0):9.1f} +{1e-3 * (e_ - s_):7.1f}  gap {1e-3 * max(0, s_ - prev):6.1f}  {n_}')
            prev = e_
        else:
            print(f'{1e-3 * (s_ - t0):9.1f} +{1e-3 * (e_ - s_):7.1f}              | {n_}')
