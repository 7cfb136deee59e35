#!/bin/bash
# Diagnostic variant of the library: tools/build_variant.sh <name> <extra hipcc flags for kernels_egnn.hip ...>
#   e.g. tools/build_variant.sh stamps1 -DCMDGEN_STAMPS=1      -> build/libcmdgen_hip_stamps1.so   (use with CMDGEN_LIB=...)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build
C=cmdgen_amd/csrc
# FILE=kernels_edge128.hip tools/build_variant.sh ... rebuilds that file instead of kernels_egnn.hip
f=${FILE:-kernels_egnn.hip}
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-function -Wno-unused-value "$@" -c $C/$f -o build/${f%.hip}_$name.o
egnn=$C/kernels_egnn.o
[ "$f" = kernels_egnn.hip ] && egnn=build/kernels_egnn_$name.o
n64o=$C/kernels_node64.o
[ "$f" = kernels_node64.hip ] && n64o=build/kernels_node64_$name.o
traino=$C/kernels_train.o
[ "$f" = kernels_train.hip ] && traino=build/kernels_train_$name.o
n16o=$C/kernels_node16w.o
[ "$f" = kernels_node16w.hip ] && n16o=build/kernels_node16w_$name.o
e128o=$C/kernels_edge128.o
[ "$f" = kernels_edge128.hip ] && e128o=build/kernels_edge128_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libcmdgen_hip_$name.so $C/cmdgen_api.o $egnn $n64o $n16o $e128o $C/kernels_ddpm.o $C/kernels_joint.o $traino $C/cmdgen_train.o
echo build/libcmdgen_hip_$name.so
