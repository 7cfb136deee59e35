#!/usr/bin/env python3
"""Per-kernel means of an SQ PMC pass (rocprofv3 --kernel-trace --pmc SQ_... -d DIR -- python3 ...), by template instance."""
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True):
    for row in csv.DictReader(open(f, newline='')):
        name = (row.get('Kernel_Name') or '').split('(')[0].replace('void ', '')
        if not name.startswith('k_'):
            continue
        s = acc[name][row['Counter_Name']]; s[0] += float(row['Counter_Value']); s[1] += 1
for name, cs in sorted(acc.items()):
    n = max(v[1] for v in cs.values())
    m = {c: v[0] / v[1] for c, v in cs.items()}
    wc = m.get('SQ_WAVE_CYCLES', 0.0)
    line = f'{name:44s} n={n:5d}'
    for c in sorted(m):
        line += f' {c.replace("SQ_", "")}={m[c]:.4g}'
    if wc:
        line += ' | of wave-cycles:' + ''.join(f' {c.replace("SQ_", "")}={m[c] / wc:.2f}' for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_LDS') if c in m)
    print(line)
