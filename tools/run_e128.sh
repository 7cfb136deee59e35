#!/bin/bash
# first contact of the 128-row edge kernels: parity tests, then per-kernel times against the 64-row kernels (same box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out
timeout -k 10 500 python -m pytest tests/test_hip_edge128.py -x -q -s > $o/r04_b_e128_tests.log 2>&1; rc=$?; tail -15 $o/r04_b_e128_tests.log
if [ $rc -ne 0 ]; then echo "tests failed rc=$rc"; exit 1; fi
for a in "256" "64 full-atom" "256 full-atom" "64"; do
  for mt in 64 128; do
    if [ $mt = 128 ]; then export CMDGEN_OPTIONS=edge_mt=128,coord_mt=128; else unset CMDGEN_OPTIONS; fi
    timeout -k 10 200 python tools/steady_profile.py $a 2>/dev/null | tail -1
  done
done > $o/r04_b_e128_profile.jsonl
cat $o/r04_b_e128_profile.jsonl
