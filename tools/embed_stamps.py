"""Per-phase cycle stamps of k_embed's full-path tiles (diagnostic build: kernels_egnn_graph.hip compiled with -DCMDGEN_STAMPS=3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _opts  # noqa: E401,F401  (CMDGEN_OPTIONS -> handle options)
import torch, numpy as np
import cmdgen_amd
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
names = ['stage weights + inputs', 'encoder layer 0', 'encoder layer 2 + time', 'embedding 33 -> 256', 'P | Q projection + stores']
print('B', B, 'full-path tiles stamped:', int(nw), 'mean cycles per wave (waves 0..3):')
for i, nm in enumerate(names):
    print(f'  {nm:28s}', [round(s[w * 8 + i] / nw) for w in range(4)])
print('  lifetime                    ', [round(s[32 + w] / nw) for w in range(4)])
