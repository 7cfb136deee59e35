#!/bin/bash
# same-box A/B of two builds of the library on the sampler: bash tools/ab_sampler_libs.sh <libA.so> <libB.so>   ("-" = the in-tree library)
# per library and shape: per-launch times of the three tile kernels and one evaluation (tools/steady_profile.py), then the 64-pocket chain
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for l in "$@"; do
  if [ "$l" = "-" ]; then unset CMDGEN_LIB; else export CMDGEN_LIB=$l; fi
  for shape in "64" "256" "64 full-atom"; do echo -n "[$l | $shape] "; timeout -k 10 100 python tools/steady_profile.py $shape 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print({k: (round(v, 1) if isinstance(v, float) else v) for k, v in d.items() if not isinstance(v, (dict, list))})"; done
  echo -n "[$l | chain 64 x 1000] "; timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --north-star-batch 0 --no-extra-shapes 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), 'pocket-steps/s', round(d['config']['us_per_denoising_step'],1), 'us/step')"
done; done
