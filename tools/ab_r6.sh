#!/bin/bash
# same-box A/B of builds of the library (round 6): bash tools/ab_r6.sh <lib.so | -> ...   ("-" = the in-tree library); two rounds, interleaved
# shapes: SHAPES="256;64 full-atom" (default), each "B [rep]"
set -u
cd "${GRAFT_REPO_ROOT:?}"
IFS=';' read -ra SH <<< "${SHAPES:-256;64 full-atom}"
for rep in 1 2; do for l in "$@"; do
  if [ "$l" = "-" ]; then unset CMDGEN_LIB; else export CMDGEN_LIB=$l; fi
  for shape in "${SH[@]}"; do echo -n "[$l] "; timeout -k 10 200 python tools/steady_ab.py $shape 2>&1 | tail -1; done
done; done
