#!/bin/bash
# option sweep on whole chains (round 6): bash tools/opt_sweep.sh "<bench.py args>" "<opts>" "<opts>" ...   ("-" = the library's own choices)
set -u
cd "${GRAFT_REPO_ROOT:?}"
args=$1; shift
for o in "$@"; do
  if [ "$o" = "-" ]; then unset CMDGEN_OPTIONS; else export CMDGEN_OPTIONS=$o; fi
  echo -n "[$o] "
  timeout -k 10 300 python bench.py $args --no-cpu-baseline --north-star-batch 0 --no-extra-shapes 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); l=d['config']['launch']; print(round(d['value']), 'pocket-steps/s', round(d['config']['us_per_denoising_step'],1), 'us/step | node', l['node_mt'], 'n64', l['node64'], 'edge', l['edge_mt'], 'coord', l['coord_mt'])"
done
