#!/bin/bash
# The library with ONE kernel file taken from another git revision (same-box A/B baselines):
#   tools/build_rev.sh <name> <rev> <file.hip> [extra hipcc flags]   -> build/libcmdgen_hip_<name>.so   (use with CMDGEN_LIB=...)
# Everything is staged under build/ (never /tmp: a stale header there shadows the real one for out-of-tree copies of csrc/).
set -e
cd "$(dirname "$0")/.."
name=$1; rev=$2; f=$3; shift 3
C=cmdgen_amd/csrc
mkdir -p build/rev_$name/cmdgen_amd/csrc build/rev_$name/include
git show $rev:$C/$f > build/rev_$name/cmdgen_amd/csrc/$f
for h in $(git ls-tree --name-only $rev $C/ | grep '\.h$'); do git show $rev:$h > build/rev_$name/$h; done
git show $rev:include/cmdgen_hip.h > build/rev_$name/include/cmdgen_hip.h
extra=$(python3 tools/file_flags.py $f)      # the flags __graft_entry__ gives this file (one table)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-function -Wno-unused-value $extra "$@" -c build/rev_$name/cmdgen_amd/csrc/$f -o build/${f%.hip}_$name.o
objs=""
for o in cmdgen_api kernels_egnn kernels_egnn_graph kernels_egnn_msg kernels_egnn_node kernels_egnn_coord kernels_egnn_graph_hx kernels_egnn_msg_hx kernels_egnn_node_hx kernels_egnn_coord_hx kernels_node64 kernels_node16w kernels_edge128 kernels_ddpm kernels_joint kernels_train cmdgen_train; do
  if [ "$o.hip" = "$f" ]; then objs="$objs build/${o}_$name.o"; else objs="$objs $C/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libcmdgen_hip_$name.so $objs
echo build/libcmdgen_hip_$name.so
