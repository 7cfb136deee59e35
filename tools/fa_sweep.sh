#!/bin/bash
run() { timeout -k 10 300 python bench.py --representation full-atom --batch 64 --timesteps 100 --steps 1 --warmup 1 --north-star-batch 0 --no-cpu-baseline --no-extra-shapes 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); pk=d['roofline']['per_kernel']
print('$1', round(d['value']), {k: round(v['avg_launch_ms']*1e3,1) for k,v in pk.items()}, d['config']['launch'])"; }
run default
CMDGEN_OPTIONS=edge_mt=32 run mt32
CMDGEN_OPTIONS=edge_mt=32,edge_wgs_per_cu=3 run mt32x3
CMDGEN_OPTIONS=edge_mt=32,edge_wgs_per_cu=4 run mt32x4
CMDGEN_OPTIONS=edge_wgs_per_cu=2 run mt64x2
run default
