#!/usr/bin/env python3
"""Which synthetic model keeps the phar points inside the pocket for a whole K=1000 chain (edges per pocket-evaluation, time)?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _opts  # noqa: E401,F401  (CMDGEN_OPTIONS -> handle options)
from cmdgen_amd import hip_backend
from cmdgen_amd.synthetic import ModelConfig, make_state_dict, make_pockets
B, T = 64, 1000
pb = make_pockets(B, 'CA', n_phar=15)
for prec, nv, gain, seed in [(0.05, (1.0, 0.5), 1e-3, 0), (0.05, (1.0, 0.5), 1.0, 0), (0.05, (1.0, 0.5), 1.0, 81), (0.1, (1.0, 0.25), 1e-3, 0), (0.01, (1.0, 1.0), 1e-3, 0)]:
    cfg = ModelConfig(timesteps=T, noise_precision=prec, norm_values=nv)
    h = hip_backend.Handle(cfg.as_dict(), 0); h.load_state_dict(make_state_dict(cfg, seed=seed, coord_gain=gain))
    h.set_layout(pb.num_nodes_phar, pb.size)
    px, poh = torch.from_numpy(pb.x).cuda(), torch.from_numpy(pb.one_hot).cuda()
    h.sample_chain(px, poh, T, noise=None, seed=1, pocket_ids=pb.pocket_index)
    torch.cuda.synchronize(); h.reset_counters(); t0 = time.perf_counter()
    out = h.sample_chain(px, poh, T, noise=None, seed=2, pocket_ids=pb.pocket_index)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    c = h.counters(); ev = c['evaluations']
    print(f'precision {prec} norm {nv} gain {gain} seed {seed}: {B * ev / dt / 1e3:.1f}k pocket-steps/s, edges/pocket-eval {c["edges"] / ev / B:.1f}, '
          f'coord edges {c["edges_phar"] / ev / B:.1f}, final max|x| {float(out[0][:, :3].abs().max()):.1f}', flush=True)
    h.close()
