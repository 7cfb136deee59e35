"""Per-phase cycle stamps of k_dgrad_tail<3> over training steps (diagnostic build:
FILE=kernels_train.hip tools/build_variant.sh tailst -DCMDGEN_TAIL_STAMPS=1; CMDGEN_LIB=build/libcmdgen_hip_tailst.so).
Every 4th workgroup reports; both lists (coordinate and message) of every block are summed.  usage: python tools/tail_stamps.py [B] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import bench_train as bt
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device('cuda:0')
cfg, model, tr = bt.build_trainer(B, 'CA', 'fp32', dev)
batches = [bt.synthetic_batch(B, B * i, dev) for i in range(2)]
if os.environ.get('TAIL_LR0'):      # diagnostic libraries with deliberately wrong arithmetic: keep the parameters (and with them the lists) where they are
    tr.lr = 0.0
for i in range(3): tr.training_step(batches[i % 2])
torch.cuda.synchronize()
tr.h.debug_stamps(True)
for i in range(steps): tr.training_step(batches[i % 2])
torch.cuda.synchronize()
s = tr.h.debug_stamps(True)
wgs = max(s[40], 1)
names = ['scalars + weight prefetch', 'GEMM (row loads, two half-K passes)', 'accumulators -> LDS', "SiLU'(pre1) on the wave's rows", 'row walk: dQ atomics, sums, dot',
         'run-end dP atomics + geometry', 'column partial sums']
print(f'B {B}: {wgs} sampled workgroups over {steps} steps')
for i, nm in enumerate(names):
    print(f'  {nm:44s}', [round(s[w * 8 + i] / wgs) for w in range(4)])
print('  lifetime', [round(s[32 + w] / wgs) for w in range(4)])
# (slots 44..47 are filled only with profiles/r05_ay_tail_gemm_stamps.patch applied: stamps inside dgrad_tile_gemm)
if any(s[44:48]):
    print('  inside the GEMM phase (wave 0): rows wait + split + barrier / GEMM, stage 0:', round(s[44] / wgs), '/', round(s[45] / wgs), ' stage 1:', round(s[46] / wgs), '/', round(s[47] / wgs))
