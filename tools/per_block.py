"""Per-block launch times from a rocprofv3 kernel trace (csv): the launches of each tile kernel in start order, folded by position in the
evaluation (n_layers launches per evaluation).   python tools/per_block.py <kernel_trace.csv> [n_layers]"""
import csv, sys, collections
L = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
fam = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    key = None
    if 'k_edge128<false>' in n or 'k_edge_msg' in n: key = 'msg'
    elif 'k_edge128<true>' in n or 'k_edge_coord' in n: key = 'coord'
    elif 'k_node' in n: key = 'node'
    if key: fam[key].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in fam.items():
    n = len(v) // L * L
    per = [sum(v[b:n:L]) / max(1, len(v[b:n:L])) / 1e3 for b in range(L)]
    print(k, 'launches', len(v), 'per block us:', ' '.join('%.1f' % x for x in per), '| mean %.1f' % (sum(per) / L))
