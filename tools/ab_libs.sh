#!/bin/bash
# Same-box A/B of SEVERAL builds: tools/ab_libs.sh "<B> [rep]" default build/libA.so build/libB.so ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
shape=$1; shift
for rep in 1 2; do
for L in "$@"; do
  if [ $L = default ]; then unset CMDGEN_LIB; else export CMDGEN_LIB=$L; fi
  echo -n "[$L] "
  timeout -k 10 200 python tools/steady_profile.py $shape 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); m=d['ms']; l=d['launch']
print('mt', l['node_mt'], l['edge_mt'], l['coord_mt'], '| msg %.1f node %.1f coord %.1f us/launch | eval %.1f us' % (m['edge_msg_ms']*200, m['node_ms']*200, m['edge_coord_ms']*200, 1e3*(m['edge_build_ms']+m['embed_ms']+m['edge_msg_ms']+m['node_ms']+m['edge_coord_ms']+m['readout_ms'])))"
done
done
