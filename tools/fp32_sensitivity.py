#!/usr/bin/env python3
"""How far does the chain's own fp32 arithmetic move the result?  The oracle (CPU restatement of the reference) run in float32 and in
float64 on the same pockets, weights and draws as golden G14's configs[1] chain (first `B` pockets): per-sample coordinate RMS between
the two = the noise floor any fp32 implementation of this chain sits on (test infrastructure; CPU only)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import ref_cpu
from helpers import NoiseTape, pocket_dict, load_golden
from cmdgen_amd.synthetic import ModelConfig, make_state_dict, make_pockets
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
G = load_golden('g14_fullsize_chains.npz'); name = 'ca_b64_K1000'
H, L, B0, R, seed, K, T, first, nseed, window = [int(v) for v in G[name + '/meta']]
cfg = ModelConfig(hidden_nf=H, n_layers=L, residue_nf=R, timesteps=T, noise_precision=0.05, norm_values=(1.0, 0.5))
sd = make_state_dict(cfg, seed=seed, coord_gain=1.0)
pb_all = make_pockets(B0, 'CA', n_phar=15, first_index=first)
gen = torch.Generator().manual_seed(nseed)
noise = torch.stack([torch.randn((B0 * 15, 11), generator=gen) for _ in range(K + 2)])[:, :B * 15].contiguous()
pb = make_pockets(B, 'CA', n_phar=15, first_index=first)
out = {}
for dt in (torch.float32, torch.float64):
    ref_cpu.FLOAT = dt
    p = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in ref_cpu.to_torch_params(sd).items()}
    class Tape:
        def __init__(s): s.i = 0
        def __call__(s, shape):
            o = noise[s.i].to(dt); s.i += 1; return o
    t0 = time.time()
    with torch.no_grad():
        x, xp, pm, qm = ref_cpu.sample_given_pocket(p, cfg.as_dict(), pocket_dict(pb), pb.num_nodes_phar, timesteps=K, noise=Tape())
    out[dt] = x.double().numpy()
    print(dt, f'{time.time() - t0:.0f} s', flush=True)
ref_cpu.FLOAT = torch.float32
d = (out[torch.float32][:, :3] - out[torch.float64][:, :3]).reshape(B, 15, 3)
e = np.sqrt((d ** 2).mean((1, 2)))
w = G[name + '/xh_phar'][:B * 15, :3].reshape(B, 15, 3)
e_ref = np.sqrt(((out[torch.float32][:, :3].reshape(B, 15, 3) - w) ** 2).mean((1, 2)))
print('oracle fp32 vs reference golden (same arithmetic: should be ~0):', np.array2string(e_ref, precision=2))
print('per-sample coordinate RMS, oracle fp32 vs oracle fp64 (A):', np.array2string(np.sort(e), precision=2))
print(f'median {np.median(e):.2e}  max {e.max():.2e}  whole-batch RMS {np.sqrt((d ** 2).mean()):.2e}; types equal: {np.array_equal(out[torch.float32][:, 3:], out[torch.float64][:, 3:])}')
