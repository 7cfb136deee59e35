"""Per-phase cycle stamps of k_node_pair (diagnostic build: FILE=kernels_node_pair.hip tools/build_variant.sh stamps4 -DCMDGEN_STAMPS=4)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from cmdgen_amd import hip_backend
from cmdgen_amd.synthetic import ModelConfig, make_state_dict, make_pockets
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = ModelConfig(residue_nf=20, timesteps=1000)
h = hip_backend.Handle(cfg.as_dict(), 0); h.load_state_dict(make_state_dict(cfg, seed=0))
pb = make_pockets(B, 'CA'); h.set_layout(pb.num_nodes_phar, pb.size)
dev = torch.device('cuda')
h.sample_chain(torch.from_numpy(pb.x).to(dev), torch.from_numpy(pb.one_hot).to(dev), 20, seed=1)
h.debug_stamps(True)
h.sample_chain(torch.from_numpy(pb.x).to(dev), torch.from_numpy(pb.one_hot).to(dev), 40, seed=2, use_graph=False)
s = h.debug_stamps(True)
nw = max(s[40], 1) / 4
names = ['launch -> tile in LDS', 'GEMM1 (W3, N-split)', 'SiLU epilogue', 'GEMM2 (W4, K-split)', 'partials out + drain', 'flag + wait for partner',
         'partner partials + h_new', 'projections + stores']
print('B', B, 'node_pair', h.query('node_pair'), 'mean cycles per wave per launch (waves 0..3):')
for i, nm in enumerate(names):
    print(f'  {nm:28s}', [round(s[w * 8 + i] / nw) for w in range(4)])
print('  lifetime                    ', [round(s[32 + w] / nw) for w in range(4)])
