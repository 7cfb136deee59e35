#!/usr/bin/env python3
"""Static instruction census of one kernel of a hipcc -S listing, basic block by basic block.

    tools/isa_count.py build/isa/edge128.s _ZN9e128_half9k_edge128ILb0E [--min 40]

Classes: MFMA (v_mfma*), VALU (other v_*; 'trans' = the quarter-rate subset: v_exp / v_rcp / v_rsq / v_sqrt / v_log / v_sin / v_cos),
DS (ds_*), VMEM (buffer_* / global_* / scratch_* / flat_*), SALU (s_* without waitcnt / barrier / nop), barriers, waits.
A block is the code between two labels (or after a branch).  Blocks with fewer than --min instructions are folded into a total line.
Dynamic counts = static counts x trip counts, which the reader supplies (the q loop of k_edge128 runs four times per tile).
"""
import re
import sys


def classify(op):
    if op.startswith('v_mfma') or op.startswith('v_smfma'):
        return 'mfma'
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('ds_'):
        return 'ds'
    if op.startswith(('buffer_', 'global_', 'scratch_', 'flat_')):
        return 'vmem'
    if op == 's_barrier':
        return 'barrier'
    if op.startswith('s_waitcnt'):
        return 'wait'
    if op.startswith(('s_cbranch', 's_branch', 's_setpc', 's_endpgm')):
        return 'branch'
    if op.startswith('s_nop'):
        return 'nop'
    if op.startswith('s_'):
        return 'salu'
    return 'other'


TRANS = ('v_exp_', 'v_rcp_', 'v_rsq_', 'v_sqrt_', 'v_log_', 'v_sin_', 'v_cos_')


def main():
    path, key = sys.argv[1], sys.argv[2]
    minn = 40
    if '--min' in sys.argv:
        minn = int(sys.argv[sys.argv.index('--min') + 1])
    detail = '--ops' in sys.argv
    lines = open(path).read().split('\n')
    start = None
    for i, l in enumerate(lines):
        if l.startswith(key) and l.rstrip().split(':')[0].startswith(key) and ':' in l:
            start = i
            break
    if start is None:
        sys.exit('kernel not found')
    blocks = []
    cur = {'label': 'entry', 'line': start, 'c': {}, 'ops': {}, 'targets': []}
    for i in range(start + 1, len(lines)):
        l = lines[i].strip()
        if l.startswith('.Lfunc_end') or l.startswith('.section') or l.startswith('.rodata'):
            break
        if not l or l.startswith(';') or l.startswith('.p2align') or l.startswith('.'):
            m = re.match(r'^(\.LBB[0-9_]+):', l)
            if m:
                blocks.append(cur)
                cur = {'label': m.group(1), 'line': i, 'c': {}, 'ops': {}, 'targets': []}
            continue
        op = l.split()[0]
        k = classify(op)
        cur['c'][k] = cur['c'].get(k, 0) + 1
        if k == 'valu':
            if op.startswith(TRANS):
                cur['c']['trans'] = cur['c'].get('trans', 0) + 1
            base = re.sub(r'_(e32|e64|dpp|sdwa)$', '', op)
            cur['ops'][base] = cur['ops'].get(base, 0) + 1
        if k == 'branch':
            t = re.search(r'(\.LBB[0-9_]+)', l)
            if t:
                cur['targets'].append(t.group(1))
    blocks.append(cur)
    tot = {}
    small = {}
    cols = ['valu', 'trans', 'mfma', 'ds', 'vmem', 'salu', 'wait', 'barrier', 'branch']
    print('%-14s %7s ' % ('block', 'line') + ' '.join('%7s' % c for c in cols) + '  -> targets')
    for b in blocks:
        n = sum(v for k, v in b['c'].items() if k != 'trans')
        for k, v in b['c'].items():
            tot[k] = tot.get(k, 0) + v
        if n < minn:
            for k, v in b['c'].items():
                small[k] = small.get(k, 0) + v
            continue
        print('%-14s %7d ' % (b['label'], b['line'] + 1) + ' '.join('%7d' % b['c'].get(c, 0) for c in cols) + '  -> ' + ','.join(b['targets']))
        if detail:
            top = sorted(b['ops'].items(), key=lambda kv: -kv[1])[:14]
            print('      ' + '  '.join('%s:%d' % kv for kv in top))
    print('%-14s %7s ' % ('(small blocks)', '') + ' '.join('%7d' % small.get(c, 0) for c in cols))
    print('%-14s %7s ' % ('total', '') + ' '.join('%7d' % tot.get(c, 0) for c in cols))


if __name__ == '__main__':
    main()
