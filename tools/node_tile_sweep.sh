#!/bin/bash
# node-kernel tile sweep (round 6): per batch size the evaluation time with the library's own choice, the 32-row plane tiles (node64=32), the 64-row
# plane tiles (node64=1) and the eight-wave 16-row tiles (node_mt=16,node64=0)
set -u
cd "${GRAFT_REPO_ROOT:?}"
for B in ${BATCHES:-48 64 80 96 112 128 160 192 224 256}; do
  for o in "-" "node64=32" "node64=1" "node_mt=16,node64=0"; do
    if [ "$o" = "-" ]; then unset CMDGEN_OPTIONS; else export CMDGEN_OPTIONS=$o; fi
    echo -n "[$o] "; timeout -k 10 120 python tools/steady_ab.py $B ${REP:-CA} 2>&1 | tail -1
  done
done
