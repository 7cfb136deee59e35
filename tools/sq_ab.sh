#!/bin/bash
# SQ counters of the full-atom edge kernel (pinned vs burst plane GEMM) and of the node kernel at 256 C-alpha pockets (k_node64 vs k_node<256,32>)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out
PMC="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"
FA="--representation full-atom --batch 64 --timesteps 20 --steps 1 --warmup 0 --north-star-batch 0 --no-cpu-baseline --no-extra-shapes"
CA="--batch 256 --timesteps 50 --steps 1 --warmup 0 --north-star-batch 0 --no-cpu-baseline --no-extra-shapes"
unset CMDGEN_LIB
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $o/sq_fa_pin -- python3 bench.py $FA > /dev/null 2>&1; echo rc=$?
export CMDGEN_LIB=build/libcmdgen_hip_burst.so
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $o/sq_fa_burst -- python3 bench.py $FA > /dev/null 2>&1; echo rc=$?
unset CMDGEN_LIB
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $o/sq_ca_n64 -- python3 bench.py $CA > /dev/null 2>&1; echo rc=$?
export CMDGEN_OPTIONS=node64=0
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $o/sq_ca_n32 -- python3 bench.py $CA > /dev/null 2>&1; echo rc=$?
unset CMDGEN_OPTIONS
for d in sq_fa_pin sq_fa_burst sq_ca_n64 sq_ca_n32; do echo "== $d"; python3 tools/sq_summary.py $o/$d | grep -E "k_edge_msg|k_node|k_edge_coord"; rm -rf $o/$d; done
