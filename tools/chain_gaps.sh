#!/bin/bash
# Idle time between consecutive kernels of the graph-replayed chain, by the kernel that FOLLOWS the gap (rocprofv3 kernel trace of a 200-step chain).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gaps -- python3 tools/concurrent_chains.py ${1:-64} 300 1 > /dev/null 2>&1
f=$(find gpurun_out/gaps -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '')[:34]) for r in csv.DictReader(open(sys.argv[1])))
# three identical graph-replayed chains (tools/concurrent_chains.py B 300 1)
rows = rows[int(len(rows) * 0.7):]          # the last of the three graph-replayed chains
gap = collections.defaultdict(lambda: [0, 0, 0]); prev_end = rows[0][1]
for s, e, n in rows[1:]:
    g = gap[n]; g[0] += max(0, s - prev_end); g[1] += 1; g[2] += e - s; prev_end = max(prev_end, e)
tot_gap = sum(v[0] for v in gap.values()); tot_busy = sum(v[2] for v in gap.values())
print('idle before / duration of (mean us), launches:')
for n, (g, c, b) in sorted(gap.items(), key=lambda kv: -kv[1][0]):
    if c > 50: print(f'  {n:36s} idle {g / c / 1e3:6.2f}  busy {b / c / 1e3:7.2f}  x {c}')
print(f'total idle {tot_gap / 1e6:.2f} ms, busy {tot_busy / 1e6:.2f} ms -> idle share {tot_gap / (tot_gap + tot_busy):.3f}')
PY
rm -rf gpurun_out/gaps
