#!/usr/bin/env python3
"""Training-step throughput (BASELINE configs[3] shape on one GPU, fp32): PharPocketDDPM training_step on synthetic
CrossDocked-shaped complexes through cmdgen_amd.training.HipTrainer.  With --gpus N (launched by torch.distributed.run)
every rank trains on its own batch and the flat gradient is all-reduced over RCCL.  Prints one JSON line.

    python tools/bench_train.py [--batch 64] [--steps 20] [--warmup 3]
"""
import argparse
import json
import os
import sys
import time
from argparse import Namespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmdgen_amd  # noqa: E402,F401
from cmdgen_amd.lightning_modules import PharPocketDDPM  # noqa: E402
from cmdgen_amd.synthetic import ModelConfig, make_pockets, make_state_dict  # noqa: E402
from cmdgen_amd.training import HipTrainer  # noqa: E402


def synthetic_batch(B, first, dev, rep='CA'):
    pb = make_pockets(B, rep, ragged=True, first_index=first)
    rng = np.random.Generator(np.random.PCG64(first))
    nl = pb.num_nodes_phar
    pm = np.repeat(np.arange(B), nl)
    com = np.stack([pb.x[pb.mask == b].mean(0) for b in range(B)])
    px = (com[pm] + rng.normal(size=(len(pm), 3)) * 2.5).astype(np.float32)
    poh = np.eye(8, dtype=np.float32)[rng.integers(0, 8, size=len(pm))]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return {'phar_coords': t(px), 'phar_one_hot': t(poh), 'num_phar_atoms': t(nl), 'phar_mask': t(pm),
            'pocket_c_alpha': t(pb.x), 'pocket_one_hot': t(pb.one_hot), 'num_pocket_nodes': t(pb.size),
            'pocket_mask': t(pb.mask),
            # host copies of the node counts, as a collate function has them before the batch is moved to the device
            'num_phar_atoms_cpu': torch.from_numpy(np.ascontiguousarray(nl)), 'num_pocket_nodes_cpu': torch.from_numpy(np.ascontiguousarray(pb.size))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--batch', type=int, default=64, help='complexes per GPU')
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--representation', default='CA', choices=['CA', 'full-atom'])
    ap.add_argument('--gemm', default='fp32', choices=['fp32', 'bf16'], help='GEMM operand precision (fp32 accumulation either way)')
    ap.add_argument('--no-pipeline', action='store_true', help='wait for every step\'s gradient norm before queueing the next step (HipTrainer.pipelined = False)')
    ap.add_argument('--cpu-baseline', action='store_true', help='also time the oracle (torch CPU autograd) on the same batch shape')
    ap.add_argument('--profile', action='store_true', help='print the per-kernel time table of 3 steps (torch.profiler)')
    a = ap.parse_args()
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('nccl')
    dev = torch.device('cuda', local)
    cfg = ModelConfig(residue_nf=20 if a.representation == 'CA' else 11)
    hp = dict(outdir='out', dataset='crossdock' if a.representation == 'CA' else 'crossdock_full', datadir='data', batch_size=a.batch, lr=1e-3,
              egnn_params=Namespace(device='cuda', edge_cutoff=6.0, joint_nf=32, hidden_nf=256, n_layers=5, attention=True,
                                    tanh=True, norm_constant=1, inv_sublayers=1, sin_embedding=False,
                                    aggregation_method='sum', normalization_factor=100),
              diffusion_params=Namespace(diffusion_steps=500, diffusion_noise_schedule='polynomial_2',
                                         diffusion_noise_precision=1e-5, diffusion_loss_type='l2', normalize_factors=[1, 4]),
              num_workers=0, augment_noise=0, augment_rotation=False, clip_grad=True, eval_epochs=50,
              eval_params=Namespace(n_eval_samples=100, eval_batch_size=100), mode='pocket_conditioning',
              node_histogram=np.ones((30, 500)), pocket_representation=a.representation)
    model = PharPocketDDPM(**hp)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(cfg, seed=0).items()}, strict=True)
    model = model.to(dev)
    tr = HipTrainer(model, gemm_dtype=a.gemm)
    tr.pipelined = not a.no_pipeline
    batches = [synthetic_batch(a.batch, 50000 + 1000 * rank + 100 * i, dev, a.representation) for i in range(4)]
    torch.manual_seed(rank)
    for i in range(a.warmup):
        tr.training_step(batches[i % 4])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    losses = []
    for i in range(a.steps):
        losses.append(tr.training_step(batches[i % 4])['loss'])          # device scalars: read after the loop
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    losses = [float(x) for x in losses]
    if world > 1:
        td = torch.tensor([dt], device=dev)
        dist.all_reduce(td, op=dist.ReduceOp.MAX)
        dt = float(td)
    if a.profile and rank == 0:
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
            for i in range(3):
                tr.training_step(batches[i % 4])
            torch.cuda.synchronize()
        rows = [(e.key, e.device_time_total, e.count) for e in prof.key_averages()
                if e.device_time_total > 0 and not e.key.startswith(('aten::', 'hip'))]      # device kernels / memcpy / memset only
        rows.sort(key=lambda r: -r[1])
        tot = sum(r[1] for r in rows)
        sys.stderr.write('device time of 3 steps: %.2f ms\n' % (tot / 1e3))
        for k, us, n in rows[:28]:
            sys.stderr.write('%8.2f ms %5.1f%% %6d  %s\n' % (us / 1e3, 100 * us / tot, n, k[:110]))
        crow = [(e.key, e.self_cpu_time_total, e.count) for e in prof.key_averages() if e.self_cpu_time_total > 0]
        crow.sort(key=lambda r: -r[1])
        sys.stderr.write('host self time of 3 steps: %.2f ms\n' % (sum(r[1] for r in crow) / 1e3))
        for k, us, n in crow[:14]:
            sys.stderr.write('%8.2f ms %6d  %s\n' % (us / 1e3, n, k[:90]))
    cpu = None
    if a.cpu_baseline and rank == 0:
        from oracle import ref_cpu          # bench-only use of the oracle, as bench.py's cpu_baseline leg
        sd = make_state_dict(cfg, seed=0)
        b0 = {k: v.cpu() for k, v in batches[0].items()}
        phar = {'x': b0['phar_coords'], 'one_hot': b0['phar_one_hot'], 'size': b0['num_phar_atoms'], 'mask': b0['phar_mask']}
        pocket = {'x': b0['pocket_c_alpha'], 'one_hot': b0['pocket_one_hot'], 'size': b0['num_pocket_nodes'], 'mask': b0['pocket_mask']}
        best = None
        for nt in (8, 16, 32, 64):
            torch.set_num_threads(nt)
            p = ref_cpu.to_torch_params(sd)
            leaves = {k: v.clone().requires_grad_(True) for k, v in p.items() if k.startswith('dynamics.')}
            p2 = dict(p); p2.update(leaves)
            opt = torch.optim.AdamW(list(leaves.values()), lr=1e-3, amsgrad=True, weight_decay=1e-12)
            ts = []
            for it in range(3):
                t0c = time.perf_counter()
                opt.zero_grad()
                t_int = torch.randint(0, 501, (a.batch, 1)).float()
                eps = [torch.randn(len(phar['mask']), 11)]
                terms = ref_cpu.ddpm_forward(p2, cfg.as_dict(), phar, pocket, t_int, eps, training=True, histogram=np.ones((30, 500)))
                nll = ref_cpu.nll_from_terms(terms, cfg.as_dict(), phar['size'], pocket['size'], training=True)
                nll.mean(0).backward()
                opt.step()
                ts.append(time.perf_counter() - t0c)
            sec = min(ts[1:])
            if best is None or sec < best[0]:
                best = (sec, nt)
        cpu = {'value': a.batch / best[0], 'unit': 'complexes/s', 'cores': best[1], 'kind': 'port',
               'sample': '1 training step (forward + autograd backward + AdamW) of the oracle on the same batch, best of a thread sweep',
               's_per_step': best[0]}
    # phase split on one more step (events around forward / backward / optimizer)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record(); tr.loss_and_grad(batches[0]); ev[1].record(); tr._allreduce(); ev[2].record(); tr.optimizer_step(); ev[3].record()
    torch.cuda.synchronize()
    if rank == 0:
        print(json.dumps({'metric': 'training complexes/s', 'value': a.batch * world * a.steps / dt, 'n_gpus': world,
                          'ms_per_step': dt / a.steps * 1e3, 'batch_per_gpu': a.batch, 'dtype': 'f32' if a.gemm == 'fp32' else 'bf16 GEMM operands, f32 accumulate/master',
                          'first_loss': losses[0], 'last_loss': losses[-1], 'cpu_baseline': cpu, 'pipelined': tr.pipelined,
                          'graph_of_last_step': {'nodes': tr.h.n_phar + tr.h.n_pocket, 'edges': tr.h.query('train_edges'),
                                                 'coord_edges': tr.h.query('train_coord_edges')},
                          'phase_ms': {'loss_and_grad': ev[0].elapsed_time(ev[1]), 'allreduce': ev[1].elapsed_time(ev[2]),
                                       'clip_and_adamw': ev[2].elapsed_time(ev[3])}}))


if __name__ == '__main__':
    main()
