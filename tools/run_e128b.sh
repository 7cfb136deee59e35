#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out
timeout -k 10 300 python -m pytest tests/test_hip_edge128.py -x -q > $o/r04_c_tests.log 2>&1; tail -2 $o/r04_c_tests.log
export CMDGEN_OPTIONS=edge_mt=128
for w in 2 1; do
  export CMDGEN_OPTIONS=edge_mt=128,e128_wgs_per_cu=$w
  echo "== $w workgroups per CU"
  CMDGEN_LIB=build/libcmdgen_hip_stamps6.so timeout -k 10 200 python tools/e128_stamps.py 64 full-atom 2>&1 | tail -9
  CMDGEN_LIB=build/libcmdgen_hip_stamps6.so timeout -k 10 200 python tools/e128_stamps.py 256 CA 2>&1 | tail -9
  timeout -k 10 200 python tools/steady_profile.py 64 full-atom 2>/dev/null | tail -1
  timeout -k 10 200 python tools/steady_profile.py 256 2>/dev/null | tail -1
done > $o/r04_c_e128_stamps.txt 2>&1
cat $o/r04_c_e128_stamps.txt
