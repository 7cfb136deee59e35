#!/bin/bash
# Same-box A/B of two builds of the library: tools/ab_lib.sh <other .so> [batch sizes...]  (default build vs CMDGEN_LIB=<other>)
other=$1; shift
for i in 1 2; do for L in default $other; do for B in ${@:-64 256}; do
  if [ $L = default ]; then unset CMDGEN_LIB; else export CMDGEN_LIB=$L; fi
  timeout -k 10 200 python bench.py --batch $B --steps 3 --warmup 1 --no-extra-shapes --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); k=d['config']['kernel_ms_one_evaluation']
print('$L', $B, round(d['value']), 'edge_msg %.4f node %.4f edge_coord %.4f ms per evaluation' % (k['edge_msg_ms'], k['node_ms'], k['edge_coord_ms']))"
done; done; done
