"""Per-phase cycle stamps of k_edge_msg over a chain (diagnostic build: tools/build_variant.sh stamps1 -DCMDGEN_STAMPS=1;
CMDGEN_LIB=build/libcmdgen_hip_stamps1.so).  Every 4th workgroup that had a tile reports.  usage: python tools/edge_stamps.py [B] [CA|full-atom] [K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _opts  # noqa: E401,F401  (CMDGEN_OPTIONS -> handle options)
import torch, numpy as np
from cmdgen_amd import hip_backend
from cmdgen_amd.synthetic import ModelConfig, make_state_dict, make_pockets
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rep = sys.argv[2] if len(sys.argv) > 2 else 'CA'
K = int(sys.argv[3]) if len(sys.argv) > 3 else 300
cfg = ModelConfig(residue_nf=20 if rep == 'CA' else 11, timesteps=1000)
h = hip_backend.Handle(cfg.as_dict(), 0); h.load_state_dict(make_state_dict(cfg, seed=0))
pb = make_pockets(B, rep); h.set_layout(pb.num_nodes_phar, pb.size)
dev = torch.device('cuda')
x, oh = torch.from_numpy(pb.x).to(dev), torch.from_numpy(pb.one_hot).to(dev)
h.sample_chain(x, oh, K, seed=1, use_graph=False)
h.debug_stamps(True); h.reset_counters()
h.sample_chain(x, oh, K, seed=1, use_graph=False)
s = h.debug_stamps(True); c = h.counters()
mt = h.query('edge_mt')
wgs, tiles = max(s[40] / 4, 1), max(s[41], 1)
names = ['indices + positions', 'build half 0 + GEMM half 0 + build half 1', 'GEMM half 1', 'barrier after GEMM', 'epilogue (SiLU -> LDS)', 'gate', 'segment sum']
print(f'B {B} {rep} K {K}: {c["edges"] / c["evaluations"]:.0f} edges per evaluation, {mt}-row tiles on {h.query("edge_grid")} workgroups; '
      f'sampled: {wgs:.0f} workgroup launches with {tiles / wgs:.2f} tiles each')
print('cycles per TILE (waves 0..3):')
tot = 0
for i, nm in enumerate(names):
    v = [s[w * 8 + i] / tiles for w in range(4)]
    tot += v[0]
    print(f'  {nm:44s}', [round(q) for q in v])
print(f'  sum {tot:.0f} per tile; lifetime of a workgroup with tiles {s[32] / wgs:.0f} cycles')
