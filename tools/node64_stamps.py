"""Per-phase cycle stamps of k_node64 (diagnostic build: FILE=kernels_node64.hip tools/build_variant.sh stamps5 -DCMDGEN_STAMPS=5 -fno-slp-vectorize)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _opts  # noqa: E401,F401  (CMDGEN_OPTIONS -> handle options)
import torch, numpy as np
from cmdgen_amd import hip_backend
from cmdgen_amd.synthetic import ModelConfig, make_state_dict, make_pockets
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rep = sys.argv[2] if len(sys.argv) > 2 else 'CA'
cfg = ModelConfig(residue_nf=20 if rep == 'CA' else 11, timesteps=1000)
h = hip_backend.Handle(cfg.as_dict(), 0); h.load_state_dict(make_state_dict(cfg, seed=0))
pb = make_pockets(B, rep); h.set_layout(pb.num_nodes_phar, pb.size)
dev = torch.device('cuda')
h.sample_chain(torch.from_numpy(pb.x).to(dev), torch.from_numpy(pb.one_hot).to(dev), 10, seed=1)
h.debug_stamps(True)
h.sample_chain(torch.from_numpy(pb.x).to(dev), torch.from_numpy(pb.one_hot).to(dev), 20, seed=2, use_graph=False)
s = h.debug_stamps(True)
nw = max(s[40], 1) / 4
names = ['launch -> tile in LDS', 'GEMM W3, agg part', 'SiLU epilogue', 'GEMM W4', 'residual + h rows out', 'projections + stores', 'GEMM W3, h part', 'agg: barrier, zero stores, / nf, split into the planes, barrier']
print('B', B, rep, 'node64', h.query('node64'), 'mean cycles per wave per WORKGROUP (waves 0..3):')
for i, nm in enumerate(names):
    print(f'  {nm:28s}', [round(s[w * 8 + i] / nw) for w in range(4)])
print('  lifetime                    ', [round(s[32 + w] / nw) for w in range(4)])
