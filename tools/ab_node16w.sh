#!/bin/bash
# same-box A/B of the eight-wave 16-row node kernel (option node16w): parity tests first, then per-kernel times and chains both ways
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_hip_split.py tests/test_hip_parity.py tests/test_hip_parity_r2.py tests/test_hip_parity_r3.py tests/test_hip_options.py -x -q -m gpu 2>&1 | tail -4 || exit 1
bash tools/run_ab_opts.sh "64" node16w=1 node16w=0
for rep in 1 2; do
for o in node16w=1 node16w=0; do
  for b in 32 64; do
    echo -n "[$o B=$b] "
    timeout -k 10 300 python bench.py --batch $b --steps 3 --warmup 1 --no-extra-shapes --no-cpu-baseline --option $o 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.1f pocket-steps/s' % d['value'], [(k['kernel'][:24], round(k['avg_us'],1)) for k in d['roofline'].get('kernels', [])][:3] if isinstance(d.get('roofline'), dict) else '')"
  done
done
done
