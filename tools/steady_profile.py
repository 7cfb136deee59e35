"""Per-kernel times of ONE evaluation at the geometry a trained model holds (phar points inside the pocket): the
`steady_state_evaluation` workload of bench.py, through cmdgen_profile_evaluation (an event pair around every launch)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _opts  # noqa: E401,F401  (CMDGEN_OPTIONS -> handle options)
import numpy as np, torch
import cmdgen_amd
from cmdgen_amd import hip_backend
from cmdgen_amd.synthetic import ModelConfig, make_state_dict, make_pockets
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
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
    kt = h.profile_evaluation(xh, xq, t)
h.reset_counters(); kt = h.profile_evaluation(xh, xq, t); c = h.counters()
print(json.dumps({'B': B, 'rep': rep, 'edges': c['edges'], 'coord_edges': c['edges_phar'], 'launch': {k: h.query(k) for k in ('node_mt', 'edge_mt', 'coord_mt', 'edge_grid', 'coord_grid')},
                  'ms': {k: round(v, 4) if isinstance(v, float) else v for k, v in kt.items()}}))
