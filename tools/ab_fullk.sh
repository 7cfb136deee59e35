#!/bin/bash
# same-box A/B of the full-K plane build of the 32-row edge tiles (option edge_fullk = 1 default / 0)
for i in 1 2; do for F in 1 0; do
  export CMDGEN_OPTIONS=edge_fullk=$F
  timeout -k 10 200 python bench.py --batch 64 --steps 3 --warmup 1 --no-extra-shapes --no-cpu-baseline --north-star-batch 0 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); pk=d['roofline']['per_kernel']
print('fullk=$F B=64 chain', round(d['value']), {k: round(v['avg_launch_ms']*1e3,2) for k,v in pk.items()}, 'steady', round(d['config']['steady_state_evaluation']['us_per_evaluation'],1))"
  for b in 64 128; do timeout -k 10 100 python tools/steady_profile.py $b 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('fullk=$F trained geometry', $b, {k: v for k, v in d['ms'].items() if k in ('edge_msg_ms','node_ms','edge_coord_ms')}, d['launch'])"; done
done; done
