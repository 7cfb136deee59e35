#!/bin/bash
# The whole library of another git revision (same-box A/B baselines that span several files):
#   tools/build_tree.sh <name> <rev>   -> build/libcmdgen_hip_<name>.so   (use with CMDGEN_LIB=...)
# The revision's csrc/ and include/ are exported under build/tree_<name>/ (nothing outside the repo) and compiled with __graft_entry__'s flags.
set -e
cd "$(dirname "$0")/.."
name=$1; rev=$2
d=build/tree_$name
rm -rf $d; mkdir -p $d
git archive $rev cmdgen_amd/csrc include | tar -x -C $d
C=$d/cmdgen_amd/csrc
objs=""
pids=""
for f in $C/*.hip; do
  b=$(basename $f .hip); extra=$(python3 tools/file_flags.py $b.hip)      # the flags __graft_entry__ gives this file (one table)
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-function -Wno-unused-value $extra -c $f -o $C/$b.o 2>/dev/null &
  pids="$pids $!"; objs="$objs $C/$b.o"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libcmdgen_hip_$name.so $objs
echo build/libcmdgen_hip_$name.so
