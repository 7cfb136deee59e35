#!/bin/bash
# same-box A/B of two source trees: the repo root vs build/wt_old (git archive <commit> | tar -x -C build/wt_old, then python __graft_entry__.py there)
for i in 1 2; do for T in . build/wt_old; do for B in 64 256; do
  ( cd $T && timeout -k 10 200 python bench.py --batch $B --steps 3 --warmup 1 --no-extra-shapes --no-cpu-baseline --north-star-batch 0 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); pk=d['roofline']['per_kernel']
print('$T', $B, round(d['value']), {k: round(v['avg_launch_ms']*1e3,2) for k,v in pk.items()})" )
done; done; done
