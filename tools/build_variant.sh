#!/bin/bash
# Diagnostic variant of the library: FILE=<one .hip of csrc/> tools/build_variant.sh <name> <extra hipcc flags for that file ...>
#   e.g. FILE=kernels_egnn_msg.hip tools/build_variant.sh stamps1 -DCMDGEN_STAMPS=1      -> build/libcmdgen_hip_stamps1.so   (use with CMDGEN_LIB=...)
# (FILE defaults to kernels_egnn_msg.hip; every other object is the in-tree one.)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build
C=cmdgen_amd/csrc
f=${FILE:-kernels_egnn_msg.hip}
extra=$(python3 tools/file_flags.py $f)      # the flags __graft_entry__ gives this file (one table); the caller's flags come after them
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-function -Wno-unused-value $extra "$@" -c $C/$f -o build/${f%.hip}_$name.o
objs=""
for o in cmdgen_api kernels_egnn kernels_egnn_graph kernels_egnn_msg kernels_egnn_node kernels_egnn_coord kernels_egnn_graph_hx kernels_egnn_msg_hx kernels_egnn_node_hx kernels_egnn_coord_hx kernels_node64 kernels_node16w kernels_edge128 kernels_ddpm kernels_joint kernels_train cmdgen_train; do
  if [ "$o.hip" = "$f" ]; then objs="$objs build/${o}_$name.o"; else objs="$objs $C/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libcmdgen_hip_$name.so $objs
echo build/libcmdgen_hip_$name.so
