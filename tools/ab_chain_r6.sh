#!/bin/bash
# same-box A/B of whole chains (round 6): bash tools/ab_chain_r6.sh "<bench.py args>" <lib.so | -> ...     two rounds, interleaved; CMDGEN_OPTIONS is passed through
set -u
cd "${GRAFT_REPO_ROOT:?}"
args=$1; shift
for rep in 1 2; do for l in "$@"; do
  if [ "$l" = "-" ]; then unset CMDGEN_LIB; else export CMDGEN_LIB=$l; fi
  echo -n "[$l${CMDGEN_OPTIONS:+ | $CMDGEN_OPTIONS}] "
  timeout -k 10 300 python bench.py $args --no-cpu-baseline --north-star-batch 0 --no-extra-shapes 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), 'pocket-steps/s', round(d['config']['us_per_denoising_step'],1), 'us/step', d['config']['launch']['node_mt'], d['config']['launch']['edge_mt'], d['config']['launch']['coord_mt'])"
done; done
