"""Do independent chains on separate HIP streams overlap on one MI355X?  One handle with B pockets against S handles with B / S pockets each
(global pocket ids kept, so every pocket's draws and result are the same), each on its own stream, driven from S host threads.
usage: python tools/concurrent_chains.py [B] [K] [only this number of streams]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _opts  # noqa: E401,F401  (CMDGEN_OPTIONS -> handle options)
import numpy as np, torch
from cmdgen_amd import hip_backend
from cmdgen_amd.synthetic import ModelConfig, make_state_dict, make_pockets

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
cfg = ModelConfig(residue_nf=20, timesteps=1000)
sd = make_state_dict(cfg, seed=0)
dev = torch.device('cuda')
results = {}
ONLY = [int(sys.argv[3])] if len(sys.argv) > 3 else (1, 2, 4)
for S in ONLY:
    parts = []
    for s in range(S):
        pb = make_pockets(B // S, 'CA', first_index=s * (B // S))
        h = hip_backend.Handle(cfg.as_dict(), 0); h.load_state_dict(sd)
        h.set_layout(pb.num_nodes_phar, pb.size)
        parts.append((h, torch.from_numpy(pb.x).to(dev), torch.from_numpy(pb.one_hot).to(dev), np.arange(s * (B // S), (s + 1) * (B // S)), torch.cuda.Stream()))
    outs = [None] * S

    def run(i, k):
        h, x, oh, ids, st = parts[i]
        with torch.cuda.stream(st):
            outs[i] = h.sample_chain(x, oh, k, seed=7, pocket_ids=ids, use_graph=os.environ.get('EAGER') is None)[0]
    for i in range(S): run(i, K)         # graph capture and step table, one handle at a time
    for k in (K, K):         # warm-up, then the timed chain
        torch.cuda.synchronize(); t0 = time.perf_counter()
        th = [threading.Thread(target=run, args=(i, k)) for i in range(S)]
        [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    results[S] = torch.cat([o.cpu() for o in outs])
    print(f'{S} stream(s) x {B // S} pockets: {dt * 1e3:8.1f} ms for K={K}  -> {B * K / dt / 1e3:7.1f}k pocket-steps/s', flush=True)
    for p in parts: p[0].close()
for S in [x for x in ONLY if x != 1 and 1 in ONLY]:
    d = (results[S][:, :3] - results[1][:, :3]).abs().max().item()
    print(f'{S} streams vs 1: max |dx| {d:.2e} (|x| up to {results[1][:, :3].abs().max().item():.0f}), types equal {bool((results[S][:, 3:] == results[1][:, 3:]).all())}')
