#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out
for rep in 1 2; do
for lib in base64 "" build/libcmdgen_hip_prio1.so build/libcmdgen_hip_prio2.so; do
  echo "== lib ${lib:-default}"
  export CMDGEN_OPTIONS=edge_mt=128,coord_mt=128
  if [ "$lib" = base64 ]; then unset CMDGEN_OPTIONS CMDGEN_LIB; elif [ -n "$lib" ]; then export CMDGEN_LIB=$lib; else unset CMDGEN_LIB; fi
  for a in "64 full-atom" "256"; do timeout -k 10 200 python tools/steady_profile.py $a 2>/dev/null | tail -1 | sed 's/"launch".*"edge_msg_ms"/ edge_msg_ms/' | cut -c1-170; done
done
done > $o/r04_d_prio.txt 2>&1
cat $o/r04_d_prio.txt
