#!/bin/bash
for i in 1 2; do for L in default build/libcmdgen_hip_burst.so; do
  if [ $L = default ]; then unset CMDGEN_LIB; else export CMDGEN_LIB=$L; fi
  timeout -k 10 300 python bench.py --representation full-atom --batch 64 --timesteps 200 --steps 1 --warmup 1 --north-star-batch 0 --no-cpu-baseline --no-extra-shapes 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); pk=d['roofline']['per_kernel']
print('$L full-atom 64', round(d['value']), {k: round(v['avg_launch_ms']*1e3,1) for k,v in pk.items()})"
  timeout -k 10 100 python tools/steady_profile.py 64 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$L trained geometry 64', d['ms'])"
  timeout -k 10 100 python tools/steady_profile.py 256 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$L trained geometry 256', d['ms'])"
done; done
