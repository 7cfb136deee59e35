"""Phase ablation of the edge-message kernel (timing-only; outputs are wrong while a phase is off)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _opts  # noqa: E401,F401  (CMDGEN_OPTIONS -> handle options)
import torch, numpy as np
import cmdgen_amd
from cmdgen_amd import hip_backend
from cmdgen_amd.synthetic import ModelConfig, make_state_dict, make_pockets
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rep = sys.argv[2] if len(sys.argv) > 2 else 'CA'
cfg = ModelConfig(residue_nf=20 if rep == 'CA' else 11, timesteps=1000)
h = hip_backend.Handle(cfg.as_dict(), 0); h.load_state_dict(make_state_dict(cfg, seed=0))
pb = make_pockets(B, rep); h.set_layout(pb.num_nodes_phar, pb.size)
dev = torch.device('cuda')
h.reset_counters()
out = h.sample_chain(torch.from_numpy(pb.x).to(dev), torch.from_numpy(pb.one_hot).to(dev), 20, seed=1)
c = h.counters(); E = c['edges'] / c['evaluations']
print('B', B, rep, 'edges/eval', E, 'ideal us at 155TF', E * 131584 / 155e12 * 1e6)
names = {0: 'full', 4: 'no gemm', 32: 'no valu role', 36: 'neither'} if os.environ.get('DUAL') else {0: 'full', 27: 'gemm only'} if os.environ.get('QUICK') else {0: 'full', 1: '-pos', 2: '-gather/build', 4: '-gemm', 8: '-segsum', 16: '-att', 6: '-build-gemm', 27: 'gemm only', 31: 'empty'}
for m, n in names.items():
    ts = [h.time_edge_kernel(2 | (m << 8), 50) * 1e3 for _ in range(3)]
    print(f'{n:16s} {min(ts):8.1f} us')
if os.environ.get('STAMPS'):
    import json
    for m, n in {0: 'full', 27: 'gemm only'}.items():
        h.debug_stamps(True)
        h.time_edge_kernel(2 | (m << 8), 10)
        s = h.debug_stamps(True)
        nw = max(s[40], 1) / 4
        names = ['idx+pos', 'build', 'gemm', 'barrier after gemm', 'epilogue', 'att', 'segsum']
        print(n, 'mean cycles per wave per launch, wave 0..3:')
        for i, nm in enumerate(names):
            print(f'  {nm:20s}', [round(s[w * 8 + i] / nw) for w in range(4)])
        print('  lifetime            ', [round(s[32 + w] / nw) for w in range(4)])
