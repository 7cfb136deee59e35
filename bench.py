#!/usr/bin/env python3
"""bench.py - denoising-steps/s of the DiffPhar sampling loop on MI355X.

One "step" of this benchmark = one pass of the hot path over one batch:
``ConditionalDDPM.sample_given_pocket`` on a batch of synthetic CrossDocked-shaped pockets
(BASELINE.json configs[1]: 64 C-alpha pockets, 1000-step DDPM chain, fp32), i.e.
timesteps+1 network evaluations per pocket.  Inputs are resident in HBM before the timed
region; noise is drawn on the device (Philox).  With N GPUs every rank runs its own shard of
pockets (weak scaling, no data-path collective); time = MAX over ranks.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--timesteps T]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).  Also reported: the algorithmic-FLOP roofline of the dominant
kernel (edge message, fp32 MFMA) measured live with HIP events, and the oracle's CPU rate on
the host cores (a baseline, not the target).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import cmdgen_amd  # noqa: E402,F401
from cmdgen_amd import hip_backend  # noqa: E402
from cmdgen_amd.synthetic import ModelConfig, make_state_dict, make_pockets  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_HBM_TBS = 8.0


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=3)
    p.add_argument('--warmup', type=int, default=1)
    p.add_argument('--batch', type=int, default=64, help='pockets per GPU')
    p.add_argument('--timesteps', type=int, default=1000, help='DDPM chain length (and T of the model)')
    p.add_argument('--representation', default='CA', choices=['CA', 'full-atom'])
    p.add_argument('--n_phar', type=int, default=15)
    p.add_argument('--no-graph', action='store_true')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--cpu-seconds', type=float, default=12.0)
    return p.parse_args()


def cpu_baseline(cfg, sd, batch, rep, n_phar, budget_s):
    """The oracle (port of the reference's eager CPU sequence) timed on the host cores on a
    bounded sample of the same workload: short chains on the same pockets.  torch's intra-op
    pool is sized by a quick calibration (small-tensor eager ops get SLOWER with hundreds of
    threads), and the thread count actually used is reported as `cores`."""
    from oracle import ref_cpu
    cpu_batch = batch if rep == 'CA' else min(batch, 8)   # the reference's N_total^2 edge build explodes beyond this
    pb = make_pockets(cpu_batch, rep, n_phar=n_phar)
    p = ref_cpu.to_torch_params(sd)
    c = cfg.as_dict()
    pocket = {'x': torch.from_numpy(pb.x), 'one_hot': torch.from_numpy(pb.one_hot),
              'size': torch.from_numpy(pb.size), 'mask': torch.from_numpy(pb.mask)}
    ncpu = os.cpu_count() or 1
    cands = sorted({n for n in (4, 8, 16, 32, 64) if n <= ncpu}) or [ncpu]    # hundreds of threads only get slower (40 s / evaluation at 256)
    best_n, best_t = cands[0], float('inf')
    with torch.no_grad():
        for n in cands:                                       # calibration: one 1-step chain (2 evaluations) each
            torch.set_num_threads(n)
            ref_cpu.sample_given_pocket(p, c, pocket, pb.num_nodes_phar, timesteps=1)
            t0 = time.perf_counter()
            ref_cpu.sample_given_pocket(p, c, pocket, pb.num_nodes_phar, timesteps=1)
            dt = time.perf_counter() - t0
            if dt < best_t:
                best_n, best_t = n, dt
            if dt > 1.3 * best_t:
                break
        torch.set_num_threads(best_n)
        K = 4
        t0 = time.perf_counter()
        evals = 0
        while True:
            ref_cpu.sample_given_pocket(p, c, pocket, pb.num_nodes_phar, timesteps=K)
            evals += K + 1
            el = time.perf_counter() - t0
            if el >= budget_s or evals >= 400:
                break
    return {'value': cpu_batch * evals / el, 'unit': 'pocket-steps/s', 'cores': int(best_n), 'kind': 'port',
            'sample': f'{evals} network evaluations (chains of {K} steps + final decode) of the same model on '
                      f'{cpu_batch} {rep} pockets, torch {torch.__version__} CPU fp32 with {best_n} of {ncpu} '
                      f'hardware threads (best of {cands}), {el:.1f} s'}


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device('cuda', local_rank if world > 1 else 0)
    n_gpus = world
    if args.gpus != world and rank == 0:
        print(f'note: --gpus {args.gpus} but WORLD_SIZE {world}; using {world}', file=sys.stderr)

    B, T, rep = args.batch, args.timesteps, args.representation
    cfg = ModelConfig(residue_nf=20 if rep == 'CA' else 11, timesteps=T)
    sd = make_state_dict(cfg, seed=0)
    h = hip_backend.Handle(cfg.as_dict(), dev.index)
    h.load_state_dict(sd)
    pb = make_pockets(B, rep, n_phar=args.n_phar, first_index=rank * B)     # shard = global pockets [rank*B, (rank+1)*B)
    h.set_layout(pb.num_nodes_phar, pb.size)
    px, poh = torch.from_numpy(pb.x).to(dev), torch.from_numpy(pb.one_hot).to(dev)
    use_graph = not args.no_graph
    stream = torch.cuda.Stream(device=dev)

    def chain(seed, graph=use_graph):
        return h.sample_chain(px, poh, T, noise=None, seed=seed, pocket_ids=pb.pocket_index, use_graph=graph)

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    with torch.cuda.stream(stream):
        for i in range(args.warmup):
            chain(1000 + i)
        fence()
        h.reset_counters()
        t0 = time.perf_counter()
        for i in range(args.steps):
            out = chain(i)
        fence()
        elapsed = time.perf_counter() - t0
        cnt = h.counters()
        st = h.chain_status()
    if dist is not None:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    evals_per_chain = T + 1
    value = n_gpus * B * evals_per_chain * args.steps / elapsed

    result = None
    if rank == 0:
        # ---- roofline of the dominant kernel: per-launch HIP events around every launch of the three MFMA
        # kernels during one more (eager) chain on the same stream; the kernel with the largest summed time is
        # the dominant one.  Algorithmic FLOP per launch: edge message 2(H^2+H) per edge,
        # node 14 H^2 per node, coord 2(H^2+H) per phar-receiver edge.
        H = cfg.hidden_nf
        L = cfg.n_layers
        with torch.cuda.stream(stream):
            h.reset_counters()
            h.set_kernel_profiling(True)
            Kp = min(T, 200)
            h.sample_chain(px, poh, Kp, noise=None, seed=7, pocket_ids=pb.pocket_index, use_graph=False)
            prof = h.kernel_profile()
            h.set_kernel_profiling(False)
            pc = h.counters()
            # whole-evaluation kernel breakdown on the final chain state
            z = out[0].clone(); z[:, 3:] = 0
            kt = h.profile_evaluation(z.contiguous(), torch.cat([out[1][:, :3], out[1][:, 3:] / cfg.norm_values[1]], 1).contiguous(),
                                      torch.full((B,), 0.5, device=dev))
        # ---- steady-state micro-benchmark: one evaluation at the geometry a TRAINED model holds -
        # phar points uniform in a 5 A ball at the pocket centre (random-init weights let the chain drift away,
        # which roughly halves the edge count).  Eager launches, 30 repetitions.
        with torch.cuda.stream(stream):
            rng = np.random.Generator(np.random.PCG64(12345 + rank))
            nl_tot = int(pb.num_nodes_phar.sum())
            pm_np = np.repeat(np.arange(B), pb.num_nodes_phar)
            com = np.stack([pb.x[pb.mask == b].mean(0) for b in range(B)])
            v = rng.normal(size=(nl_tot, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
            xin = (com[pm_np] + v * 5.0 * np.cbrt(rng.uniform(size=(nl_tot, 1)))).astype(np.float32)
            xh_in = torch.from_numpy(np.concatenate([xin, rng.normal(size=(nl_tot, cfg.phar_nf)).astype(np.float32)], 1)).to(dev)
            xq_in = torch.from_numpy(np.concatenate([pb.x, pb.one_hot / cfg.norm_values[1]], 1).astype(np.float32)).to(dev)
            t_in = torch.full((B,), 0.5, device=dev)
            for _ in range(3):
                h.dynamics_forward(xh_in, xq_in, t_in, want_pocket=False)
            torch.cuda.synchronize(dev)
            h.reset_counters()
            t1 = time.perf_counter()
            reps = 30
            for _ in range(reps):
                h.dynamics_forward(xh_in, xq_in, t_in, want_pocket=False)
            torch.cuda.synchronize(dev)
            dt_micro = (time.perf_counter() - t1) / reps
            mc = h.counters()
        steady = {'us_per_evaluation': 1e6 * dt_micro, 'pocket_evaluations_per_s': B / dt_micro,
                  'edges_per_pocket': mc['edges'] / max(mc['evaluations'], 1) / B,
                  'coord_edges_per_pocket': mc['edges_phar'] / max(mc['evaluations'], 1) / B,
                  'alg_tflops': (L * (2.0 * (H * H + H) * (mc['edges'] + mc['edges_phar']) + 917504.0 * (H / 256.0) ** 2 * mc['nodes'])
                                 + 33792.0 * (H / 256.0) * mc['nodes']) / max(mc['evaluations'], 1) / dt_micro / 1e12}
        ev = max(pc['evaluations'], 1)
        units = {'edge_msg': pc['edges'] / ev, 'node': pc['nodes'] / ev, 'edge_coord': pc['edges_phar'] / ev}
        flop_unit = {'edge_msg': 2.0 * (H * H + H), 'node': 14.0 * H * H, 'edge_coord': 2.0 * (H * H + H)}
        kname = {'edge_msg': 'k_edge_msg (GCL.edge_model + attention + segment sum)',
                 'node': 'k_node (GCL.node_model + P/Q projections for the coord MLP and the next block)',
                 'edge_coord': 'k_edge_coord (EquivariantUpdate.coord_model)'}
        per_kernel = {}
        for k, (ms_k, n_k) in prof.items():
            avg = ms_k / max(n_k, 1)
            fl = flop_unit[k] * units[k]
            per_kernel[k] = {'total_ms': ms_k, 'launches': n_k, 'avg_launch_ms': avg, 'flop_per_launch': fl,
                             'tflops': (fl / (avg * 1e-3) / 1e12) if avg > 0 else 0.0}
        dom = max(per_kernel, key=lambda k: per_kernel[k]['total_ms'])
        achieved = per_kernel[dom]['tflops']
        avg_ms, launches, flop_per_launch = per_kernel[dom]['avg_launch_ms'], per_kernel[dom]['launches'], per_kernel[dom]['flop_per_launch']
        # whole-job algorithmic FLOP (SURVEY 8d F_alg) for the timed region
        f_alg = L * (2.0 * (H * H + H) * (cnt['edges'] + cnt['edges_phar']) + 917504.0 * (H / 256.0) ** 2 * cnt['nodes']) \
            + 33792.0 * (H / 256.0) * cnt['nodes']
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'kernel_traffic.json')
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get('workload_batch') == B and tj.get('representation') == rep:
                    traffic = tj.get('hbm_bytes_per_launch', {}).get(dom) if isinstance(tj.get('hbm_bytes_per_launch'), dict) else None
            except Exception:
                traffic = None
        result = {
            'metric': 'denoising steps/sec', 'value': value, 'unit': 'pocket-steps/s',
            'n_gpus': n_gpus, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {
                'workload': f'BASELINE.json configs[1]: batch {B} CrossDocked-shaped {rep} pockets per GPU '
                            f'(Np={int(pb.size[0])}, Nl={args.n_phar}), {T}-step DDPM sampling '
                            f'(sample_given_pocket: {evals_per_chain} network evaluations per pocket), fp32; '
                            f'one bench step = one such chain',
                'pockets_per_gpu': B, 'timesteps': T, 'representation': rep,
                'model': f'EGNN denoiser hidden_nf={H} n_layers={L} joint_nf={cfg.joint_nf} cutoff={cfg.edge_cutoff}, '
                         f'random-init weights (seed 0)',
                'hip_graph': use_graph, 'noise': 'on-device Philox4x32-10',
                'us_per_denoising_step': 1e6 * elapsed / (args.steps * evals_per_chain),
                'edges_per_pocket_eval': cnt['edges'] / max(cnt['evaluations'], 1) / B,
                'phar_edges_per_pocket_eval': cnt['edges_phar'] / max(cnt['evaluations'], 1) / B,
                'whole_step_alg_tflops': f_alg / elapsed / 1e12,
                'chain_status': st,
                'kernel_ms_one_evaluation': kt,
                'steady_state_evaluation': steady,
            },
            'roofline': {
                'bound': 'mfma', 'kernel': kname[dom],
                'achieved': achieved, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                'frac': achieved / PEAK_FP32_MFMA_TFLOPS, 'traffic': traffic,
                'flop_per_launch': flop_per_launch, 'avg_launch_ms': avg_ms, 'launches_timed': launches,
                'units_per_launch': units[dom],
                'whole_job_frac': f_alg / elapsed / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                # the north-star also asks for the HBM view: PMC bytes per launch / launch time / 8 TB/s (not the binding roofline)
                'hbm_frac_from_pmc': (traffic / (avg_ms * 1e-3) / (PEAK_HBM_TBS * 1e12)) if traffic else None,
                'per_kernel': {k: {kk: v[kk] for kk in ('total_ms', 'avg_launch_ms', 'tflops')} | {'frac': v['tflops'] / PEAK_FP32_MFMA_TFLOPS}
                               for k, v in per_kernel.items()},
            },
        }
        if not args.no_cpu_baseline and n_gpus == 1:      # timed on rank 0 at N=1 only
            result['cpu_baseline'] = cpu_baseline(cfg, sd, B, rep, args.n_phar, args.cpu_seconds)
            result['config']['gpu_over_cpu'] = value / result['cpu_baseline']['value']
        else:
            result['cpu_baseline'] = None
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == '__main__':
    main()
