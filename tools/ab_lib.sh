#!/bin/bash
# Same-box A/B of two builds of the library: tools/ab_lib.sh <other .so> [batch sizes...]  (default build vs CMDGEN_LIB=<other>)
other=$1; shift
for i in 1 2; do for L in default $other; do for B in ${@:-64 256}; do
  if [ $L = default ]; then unset CMDGEN_LIB; else export CMDGEN_LIB=$L; fi
  timeout -k 10 200 python bench.py --batch $B --steps 3 --warmup 1 --no-extra-shapes --no-cpu-baseline --north-star-batch 0 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); pk=d['roofline']['per_kernel']
print('$L', $B, round(d['value']), 'us per launch over the chain:', {k: round(v['avg_launch_ms'] * 1e3, 2) for k, v in pk.items()})"
done; done; done
