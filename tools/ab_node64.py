"""A/B of the 64-row planes node kernel against the default node kernel over batch sizes: chain throughput (graph-replayed K steps)
with option node64 = 0 / 1.  usage: python tools/ab_node64.py [CA|full] B [B ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _opts  # noqa: E401,F401  (CMDGEN_OPTIONS -> handle options)
import torch
from cmdgen_amd import hip_backend
from cmdgen_amd.synthetic import ModelConfig, make_state_dict, make_pockets
rep = sys.argv[1]
K = 50
dev = torch.device('cuda')
for B in [int(a) for a in sys.argv[2:]]:
    cfg = ModelConfig(residue_nf=20 if rep == 'CA' else 11, timesteps=1000)
    pb = make_pockets(B, 'CA' if rep == 'CA' else 'full-atom')
    x, oh = torch.from_numpy(pb.x).to(dev), torch.from_numpy(pb.one_hot).to(dev)
    row = []
    for v in ('0', '1'):
        h = hip_backend.Handle(cfg.as_dict(), 0); h.load_state_dict(make_state_dict(cfg, seed=0))
        h.set_option('node64', int(v))
        h.set_layout(pb.num_nodes_phar, pb.size)
        h.sample_chain(x, oh, K, seed=1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): h.sample_chain(x, oh, K, seed=2)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        row.append((h.query('node64'), B * K / dt))
        del h
    n = int(sum(pb.num_nodes_phar) + sum(pb.size))
    print(f'{rep} B={B} N={n} tiles64={(n + 63) // 64}: 32-row {row[0][1] / 1e3:.1f}k  64-row {row[1][1] / 1e3:.1f}k pocket-steps/s  ({row[1][1] / row[0][1]:.3f}x) flags {row[0][0]}{row[1][0]}', flush=True)
