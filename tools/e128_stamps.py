"""Per-phase cycle stamps of k_edge128 (diagnostic build: FILE=kernels_edge128.hip tools/build_variant.sh stamps6 -DCMDGEN_STAMPS=6 -fno-slp-vectorize;
CMDGEN_LIB=build/libcmdgen_hip_stamps6.so) over repeated evaluations at the geometry a trained model holds.  usage: python tools/e128_stamps.py [B] [CA|full-atom]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _opts  # noqa: E401,F401  (CMDGEN_OPTIONS -> handle options)
import numpy as np, torch
from cmdgen_amd import hip_backend
from cmdgen_amd.synthetic import ModelConfig, make_state_dict, make_pockets
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rep = sys.argv[2] if len(sys.argv) > 2 else 'CA'
cfg = ModelConfig(residue_nf=20 if rep == 'CA' else 11, timesteps=1000)
h = hip_backend.Handle(cfg.as_dict(), 0); h.load_state_dict(make_state_dict(cfg, seed=0))
pb = make_pockets(B, rep); h.set_layout(pb.num_nodes_phar, pb.size)
dev = torch.device('cuda')
rng = np.random.Generator(np.random.PCG64(12345))
nl = int(pb.num_nodes_phar.sum())
pm = np.repeat(np.arange(B), pb.num_nodes_phar)
com = np.stack([pb.x[pb.mask == b].mean(0) for b in range(B)])
v = rng.normal(size=(nl, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
xin = (com[pm] + v * 5.0 * np.cbrt(rng.uniform(size=(nl, 1)))).astype(np.float32)
xh = torch.from_numpy(np.concatenate([xin, rng.normal(size=(nl, cfg.phar_nf)).astype(np.float32)], 1)).to(dev)
xq = torch.from_numpy(np.concatenate([pb.x, pb.one_hot / cfg.norm_values[1]], 1).astype(np.float32)).to(dev)
t = torch.full((B,), 0.5, device=dev)
for _ in range(3):
    h.dynamics_forward(xh, xq, t)
torch.cuda.synchronize()
h.debug_stamps(True)
N = 10
for _ in range(N):
    h.dynamics_forward(xh, xq, t)
torch.cuda.synchronize()
s = h.debug_stamps(True)
wgs, tiles = max(s[40] / 4, 1), max(s[41], 1)
MODE9 = os.environ.get('E128_STAMP_MODE') == '9'
names = ['index phase: work (wave 0; the others idle)', 'index phase: wait at its barrier', 'builds + GEMMs (all four quarters, barriers included)', 'SiLU + row dot + partials to LDS: work', 'partials: wait at the barrier', 'gates (every wave for itself)', 'gate multiply + lane swap + ordered scan + stores: work', 'end-of-tile barrier: wait'] if MODE9 else ['index phase', 'builds: wait at their barrier', 'GEMMs: wait at their barrier', 'SiLU + row dot + exchange (+ barrier)', 'segment sum / stores + end-of-tile barrier', 'gates (every wave for itself)', 'builds: work', 'GEMMs: work']
print(f'B {B} {rep}: edge_mt {h.query("edge_mt")} coord_mt {h.query("coord_mt")}; sampled {wgs:.0f} workgroup launches (message kernel; -DCMDGEN_STAMP_COORD=1 builds record the coordinate kernel), {tiles / wgs:.2f} tiles each')
print('cycles per TILE (waves 0..3):')
tot = 0
for i, nm in enumerate(names):
    vv = [s[w * 8 + i] / tiles for w in range(4)]
    tot += vv[0]
    print(f'  {nm:34s}', [round(q) for q in vv])
print(f'  sum {tot:.0f} per tile; lifetime of a workgroup with tiles {s[32] / wgs:.0f} cycles')
