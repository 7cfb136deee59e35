"""Largest-shape sanity of the sampler (not a test: seconds of GPU time, GBs of workspace): batches far above BASELINE's, sized for the 288 GB of an
MI355X - one evaluation of 1024 full-atom pockets (390k nodes, ~14M edges) against the same pockets evaluated in four batches of 256, and a
short chain of 4096 C-alpha pockets (242k nodes) with its in-loop checks."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from cmdgen_amd import hip_backend
from cmdgen_amd.synthetic import ModelConfig, make_state_dict, make_pockets
from bench import bounded_config
from test_hip_properties import eval_inputs, handle_for, forward
out = {}
cfg = ModelConfig(residue_nf=11, timesteps=1000); sd = make_state_dict(cfg, seed=0)
pb = make_pockets(1024, 'full-atom'); xh, xq, t = eval_inputs(pb, cfg)
h = handle_for(cfg, sd, pb)
torch.cuda.synchronize(); t0 = time.perf_counter(); e = forward(h, xh, xq, t); dt = time.perf_counter() - t0
c = h.counters(); h.close()
ps, pe = np.concatenate([[0], np.cumsum(pb.num_nodes_phar)]), np.concatenate([[0], np.cumsum(pb.size)])
worst = 0.0
for q in range(4):
    lo, hi = 256 * q, 256 * (q + 1)
    h2 = hip_backend.Handle(cfg.as_dict(), 0); h2.load_state_dict(sd); h2.set_layout(pb.num_nodes_phar[lo:hi], pb.size[lo:hi])
    e2 = forward(h2, np.ascontiguousarray(xh[ps[lo]:ps[hi]]), np.ascontiguousarray(xq[pe[lo]:pe[hi]]), np.ascontiguousarray(t[lo:hi])); h2.close()
    worst = max(worst, float(np.abs(e2 - e[ps[lo]:ps[hi]]).max()))
out['fullatom_1024'] = {'nodes': int(pb.size.sum() + pb.num_nodes_phar.sum()), 'edges': int(c['edges']), 'finite': bool(np.isfinite(e).all()),
                        'first_call_s': round(dt, 3), 'max_abs_diff_vs_four_batches_of_256': worst, 'max_abs_eps': float(np.abs(e).max())}
cfg = bounded_config(20, 1000); sd = make_state_dict(cfg, seed=0)
pb = make_pockets(4096, 'CA'); h = handle_for(cfg, sd, pb)
dev = torch.device('cuda'); K = 20
t0 = time.perf_counter()
x, xp, _ = h.sample_chain(torch.from_numpy(pb.x).to(dev), torch.from_numpy(pb.one_hot).to(dev), K, seed=3, pocket_ids=pb.pocket_index)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
st = h.chain_status(); c = h.counters(); h.close()
x = x.cpu().numpy()
out['ca_4096_chain'] = {'nodes': int(pb.size.sum() + pb.num_nodes_phar.sum()), 'steps': K, 'seconds_incl_graph_capture': round(dt, 3), 'finite': bool(np.isfinite(x).all()),
                        'one_hot_rows_valid': bool((x[:, 3:].sum(1) == 1).all()), 'status': st, 'edges_per_pocket_eval': c['edges'] / c['evaluations'] / 4096}
print(json.dumps(out, indent=1))
