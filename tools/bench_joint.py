#!/usr/bin/env python3
"""Throughput of the JOINT model's RePaint inpainting chain (EnVariationalDiffusion.inpaint, every pocket
node fixed - the generate_phars call of mode 'joint') on synthetic CrossDocked-shaped pockets.
Not the headline metric (bench.py measures the shipped conditional sampler); numbers go to profiles/.

    python tools/bench_joint.py [--batch 64] [--timesteps 1000] [--resamplings 1] [--jump 1]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _opts  # noqa: E401,F401  (CMDGEN_OPTIONS -> handle options)
import cmdgen_amd  # noqa: E402,F401
from cmdgen_amd import hip_backend  # noqa: E402
from cmdgen_amd.synthetic import ModelConfig, make_pockets, make_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--timesteps', type=int, default=1000)
    ap.add_argument('--resamplings', type=int, default=1)
    ap.add_argument('--jump', type=int, default=1)
    ap.add_argument('--n_phar', type=int, default=15)
    ap.add_argument('--reps', type=int, default=2)
    a = ap.parse_args()
    cfg = ModelConfig(timesteps=a.timesteps, update_pocket_coords=True)
    h = hip_backend.Handle(cfg.as_dict(), 0)
    h.load_state_dict(make_state_dict(cfg, seed=0))
    pb = make_pockets(a.batch, 'CA', n_phar=a.n_phar)
    nl = pb.num_nodes_phar
    Nl, Np = int(nl.sum()), len(pb.mask)
    h.set_layout(nl, pb.size)
    d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    phar = (torch.zeros(Nl, 3).cuda(), torch.zeros(Nl, 8).cuda())
    pocket = (d(pb.x), d(pb.one_hot))
    fp, fq = torch.zeros(Nl).cuda(), torch.ones(Np).cuda()
    n_steps, n_draws = h.joint_plan(a.timesteps, a.resamplings, a.jump, True)

    def chain(graph=True, K=a.timesteps):
        return h.joint_chain(K, phar=phar, pocket=pocket, phar_fixed=fp, pocket_fixed=fq, resamplings=a.resamplings,
                             jump_length=a.jump, seed=1, pocket_ids=pb.pocket_index, use_graph=graph)
    chain(); torch.cuda.synchronize()
    h.reset_counters()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        chain()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    st = h.chain_status()
    cn = h.counters()
    # per-kernel event timing on a short eager chain
    Kp = min(100, a.timesteps)
    chain(False, Kp); torch.cuda.synchronize()
    h.set_kernel_profiling(True)
    chain(False, Kp)
    prof = h.kernel_profile()
    h.set_kernel_profiling(False)
    print(json.dumps({
        'workload': f'joint inpaint, B={a.batch} CA pockets (Np=44, Nl={a.n_phar}), K={a.timesteps}, '
                    f'resamplings={a.resamplings}, jump_length={a.jump}',
        'denoising_steps': n_steps, 'combined_draws': n_draws, 'chain_s': dt,
        'pocket_steps_per_s': a.batch * n_steps / dt, 'us_per_step': dt / (n_steps + 1) * 1e6,
        'edges_per_evaluation': cn['edges'] / max(1, cn['evaluations']), 'coordinate_edges_per_evaluation': cn['edges_phar'] / max(1, cn['evaluations']),
        'tiles': {k: h.query(k) for k in ('node_mt', 'edge_mt', 'coord_mt')},
        'status': st, 'kernel_profile': prof}))


if __name__ == '__main__':
    main()
