#!/bin/bash
# same-box A/B of the readout inside the step kernel (option fused_readout): tests first, then chains both ways, twice
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_hip_options.py tests/test_hip_parity.py tests/test_hip_parity_r3.py -x -q -m gpu 2>&1 | tail -5 || exit 1
for rep in 1 2; do
for o in fused_readout=1 fused_readout=0; do
  for b in 64 256; do
    echo -n "[$o B=$b] "
    timeout -k 10 300 python bench.py --batch $b --steps 3 --warmup 1 --no-extra-shapes --no-cpu-baseline --option $o 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.1f pocket-steps/s, %.1f us/step' % (d['value'], 1e3*d['ms_per_step']/d['config'].get('denoising_steps',1000)))"
  done
done
done
