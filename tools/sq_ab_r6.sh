#!/bin/bash
# SQ counters of the 128-row edge kernels, same box, one pass per library (round 6):  bash tools/sq_ab_r6.sh <lib.so | -> ...
#   workload: 256 C-alpha pockets, 50-step chain of the bounded-schedule model (the north-star batch); per kernel: vector instructions per MFMA,
#   the matrix pipe's busy share of launch x SIMDs (SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM-free estimate: SQ_BUSY_CYCLES per SE is not
#   used - the launch time in cycles comes from the kernel trace at the 2.1 GHz these launches hold), wait share of wave cycles.
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
o=gpurun_out
PMC="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS"
ARGS="${SQ_ARGS:---batch 256 --timesteps 50 --steps 1 --warmup 0 --north-star-batch 0 --no-cpu-baseline --no-extra-shapes}"
i=0
for l in "$@"; do
  i=$((i + 1))
  if [ "$l" = "-" ]; then unset CMDGEN_LIB; else export CMDGEN_LIB=$l; fi
  rm -rf $o/sq_r6_$i
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $o/sq_r6_$i -- python3 bench.py $ARGS > /dev/null 2> $o/sq_r6_$i.err
  echo "== $l (rc=$?)"
  python3 - $o/sq_r6_$i <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
dur = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f, newline='')):
        n = r.get('Kernel_Name') or ''
        key = 'k_edge128<msg>' if 'k_edge128<false>' in n else 'k_edge128<coord>' if 'k_edge128<true>' in n else 'k_node64' if 'k_node64' in n else None
        if key:
            s = acc[key][r['Counter_Name']]; s[0] += float(r['Counter_Value']); s[1] += 1
            if 'Start_Timestamp' in r and r['Counter_Name'] == 'SQ_WAVE_CYCLES':
                d = dur[key]; d[0] += float(r['End_Timestamp']) - float(r['Start_Timestamp']); d[1] += 1
for k, cs in sorted(acc.items()):
    m = {c: v[0] / max(v[1], 1) for c, v in cs.items()}
    us = dur[k][0] / max(dur[k][1], 1) / 1e3
    cyc = us * 2.1e3 * 1024            # launch x SIMDs at 2.1 GHz (profiled passes run ~1.9-2.0: the share below is a lower bound)
    print('%-18s n=%4d  %.1f us/launch (profiled)  VALU/MFMA %.2f  (VALU %.3gM, MFMA %.3gM)  MFMA busy %.2f of launch x SIMDs  wait %.2f, wait-inst %.2f, VALU-active %.2f of wave cycles' % (
        k, max(v[1] for v in cs.values()), us, m['SQ_INSTS_VALU'] / max(m['SQ_INSTS_MFMA'], 1), m['SQ_INSTS_VALU'] / 1e6, m['SQ_INSTS_MFMA'] / 1e6,
        m['SQ_VALU_MFMA_BUSY_CYCLES'] / max(cyc, 1), m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES'], m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES'], m['SQ_ACTIVE_INST_VALU'] / m['SQ_WAVE_CYCLES']))
PY
  rm -rf $o/sq_r6_$i
done
