"""Same-box A/B helper (round 6): per-launch times of the three tile kernels (median of 9 event-timed evaluations at the trained geometry)
and the graph-replayed time of one evaluation, for the library CMDGEN_LIB names (default: the in-tree one).
    python tools/steady_ab.py 256 [CA|full-atom]   ->  one line"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _opts  # noqa: E401,F401  (CMDGEN_OPTIONS -> handle options)
import numpy as np, torch
import cmdgen_amd  # noqa: F401
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
    h.profile_evaluation(xh, xq, t)
runs = [h.profile_evaluation(xh, xq, t) for _ in range(9)]
med = lambda k: float(np.median([r[k] for r in runs]))
L = cfg.n_layers
ms = h.time_evaluation(xh, xq, t, graph_len=10, replays=10 if rep == 'CA' else 3)
print('B %d %s mt %d/%d/%d | msg %.1f node %.1f coord %.1f us/launch | embed %.1f readout %.1f | graph-replayed evaluation %.1f us' % (
    B, rep, h.query('node_mt'), h.query('edge_mt'), h.query('coord_mt'), med('edge_msg_ms') * 1e3 / L, med('node_ms') * 1e3 / L, med('edge_coord_ms') * 1e3 / L,
    med('embed_ms') * 1e3, med('readout_ms') * 1e3, ms * 1e3))
