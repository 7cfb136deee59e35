#!/bin/bash
# Diagnostic variant of the library: tools/build_variant.sh <name> <extra hipcc flags for kernels_egnn.hip ...>
#   e.g. tools/build_variant.sh stamps1 -DCMDGEN_STAMPS=1      -> build/libcmdgen_hip_stamps1.so   (use with CMDGEN_LIB=...)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build
C=cmdgen_amd/csrc
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-function -Wno-unused-value "$@" -c $C/kernels_egnn.hip -o build/kernels_egnn_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libcmdgen_hip_$name.so $C/cmdgen_api.o build/kernels_egnn_$name.o $C/kernels_ddpm.o $C/kernels_joint.o $C/kernels_train.o $C/cmdgen_train.o
echo build/libcmdgen_hip_$name.so
