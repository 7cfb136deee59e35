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
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _opts  # noqa: E402,F401   (CMDGEN_OPTIONS="wgrad_stream=0,..." -> every new handle)
from cmdgen_amd.lightning_modules import PharPocketDDPM  # noqa: E402
from cmdgen_amd.synthetic import ModelConfig, make_pockets, make_state_dict  # noqa: E402
from cmdgen_amd.collectives import host_barrier, wait_collective  # noqa: E402
from cmdgen_amd.training import HipTrainer  # noqa: E402


def synthetic_batch(B, first, dev, rep='CA'):
    from cmdgen_amd.synthetic import make_training_batch
    nb = make_training_batch(B, first, rep)
    return {k: (torch.from_numpy(np.ascontiguousarray(v)) if k.endswith('_cpu') else torch.from_numpy(np.ascontiguousarray(v)).to(dev))
            for k, v in nb.items()}


def build_trainer(batch, representation, gemm, dev, pipelined=True, mode='pocket_conditioning'):
    """PharPocketDDPM with the shipped hyper-parameters (configs/crossdocked_ca_cond.yml) and seeded weights -> HipTrainer"""
    cfg = ModelConfig(residue_nf=20 if representation == 'CA' else 11, update_pocket_coords=(mode == 'joint'))
    hp = dict(outdir='out', dataset='crossdock' if representation == 'CA' else 'crossdock_full', datadir='data', batch_size=batch, lr=1e-3,
              egnn_params=Namespace(device='cuda', edge_cutoff=6.0, joint_nf=32, hidden_nf=256, n_layers=5, attention=True,
                                    tanh=True, norm_constant=1, inv_sublayers=1, sin_embedding=False,
                                    aggregation_method='sum', normalization_factor=100),
              diffusion_params=Namespace(diffusion_steps=500, diffusion_noise_schedule='polynomial_2',
                                         diffusion_noise_precision=1e-5, diffusion_loss_type='l2', normalize_factors=[1, 4]),
              num_workers=0, augment_noise=0, augment_rotation=False, clip_grad=True, eval_epochs=50,
              eval_params=Namespace(n_eval_samples=100, eval_batch_size=100), mode=mode,
              node_histogram=np.ones((30, 500)), pocket_representation=representation)
    model = PharPocketDDPM(**hp)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(cfg, seed=0).items()}, strict=True)
    model = model.to(dev)
    tr = HipTrainer(model, gemm_dtype=gemm)
    tr.pipelined = pipelined
    return cfg, model, tr


def time_training(tr, batches, steps, warmup, dev, dist=None):
    """-> (seconds for `steps` training steps after `warmup`, losses); MAX over ranks when distributed"""
    for i in range(warmup):
        tr.training_step(batches[i % len(batches)])
    torch.cuda.synchronize(dev)
    if dist is not None:
        host_barrier(device=dev)
    t0 = time.perf_counter()
    losses = []
    for i in range(steps):
        losses.append(tr.training_step(batches[i % len(batches)])['loss'])          # device scalars: read after the loop
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    if dist is not None:
        td = torch.tensor([dt], device=dev)
        wait_collective(dist.all_reduce(td, op=dist.ReduceOp.MAX, async_op=True))
        dt = float(td)
    return dt, [float(x) for x in losses]


def exposed_allreduce_ms(tr, batch, dev, reps=5):
    """What the gradient all-reduce costs a step: the same step with the chunked all-reduce overlapped with the backward pass
    (the default), with one flat all-reduce after the pass, and with no all-reduce at all (group of one would skip it: the
    backward pass alone).  -> dict of mean ms of loss_and_grad + _allreduce."""
    out = {}

    def run(label, overlap, skip):
        tr.overlap_allreduce = overlap
        ts = []
        for _ in range(reps + 1):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            tr.loss_and_grad(batch)
            if not skip:
                tr._allreduce()
            else:
                for w in tr._pending:
                    w.wait()
                tr._pending = []
            torch.cuda.synchronize(dev)
            ts.append(time.perf_counter() - t0)
        out[label] = 1e3 * float(np.mean(ts[1:]))
    run('overlapped_chunks_ms', True, False)
    run('flat_after_backward_ms', False, False)
    world, forced = tr._world, tr.force_collectives
    tr._world = lambda: 1                      # the pass without any collective
    tr.force_collectives = False
    run('no_allreduce_ms', False, True)
    tr._world, tr.force_collectives = world, forced
    tr.overlap_allreduce = True
    out['exposed_ms'] = out['overlapped_chunks_ms'] - out['no_allreduce_ms']
    out['hidden_ms'] = out['flat_after_backward_ms'] - out['overlapped_chunks_ms']
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--batch', type=int, default=64, help='complexes per GPU')
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--representation', default='CA', choices=['CA', 'full-atom'])
    ap.add_argument('--gemm', default='fp32', choices=['fp32', 'bf16'], help='GEMM operand precision (fp32 accumulation either way)')
    ap.add_argument('--mode', default='pocket_conditioning', choices=['pocket_conditioning', 'pocket_conditioning_simple', 'joint'])
    ap.add_argument('--unfused-loss', action='store_true', help='loss side with tensor operations (HipTrainer.fused_loss = False)')
    ap.add_argument('--no-pipeline', action='store_true', help='wait for every step\'s gradient norm before queueing the next step (HipTrainer.pipelined = False)')
    ap.add_argument('--cpu-baseline', action='store_true', help='also time the oracle (torch CPU autograd) on the same batch shape')
    ap.add_argument('--gloo', action='store_true', help='rendezvous over gloo (rehearsal of the multi-rank path on a one-GPU box: all ranks on cuda:0)')
    ap.add_argument('--force-dist', action='store_true', help='initialise the process group (RCCL) even with one rank and run the staged backward + chunked all-reduce through it (tests/test_hip_rccl.py)')
    ap.add_argument('--torch-stream', action='store_true', help='run on a torch side stream instead of the legacy default stream')
    ap.add_argument('--profile', action='store_true', help='print the per-kernel time table of 3 steps (torch.profiler)')
    a = ap.parse_args()
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if a.gloo:
        local = 0
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or a.force_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo' if a.gloo else 'nccl', **({} if a.gloo else {'device_id': torch.device('cuda', local)}))
    dev = torch.device('cuda', local)
    if a.torch_stream:
        torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    cfg, model, tr = build_trainer(a.batch, a.representation, a.gemm, dev, pipelined=not a.no_pipeline, mode=a.mode)
    tr.fused_loss = not a.unfused_loss
    tr.force_collectives = bool(a.force_dist)
    if dist is not None:
        tr.broadcast_state(0)
    batches = [synthetic_batch(a.batch, 50000 + 1000 * rank + 100 * i, dev, a.representation) for i in range(4)]
    torch.manual_seed(rank)
    dt, losses = time_training(tr, batches, a.steps, a.warmup, dev, dist)
    allreduce = exposed_allreduce_ms(tr, batches[0], dev) if dist is not None else None
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
                          'collectives': (dist.get_backend() if dist is not None else None),
                          'ms_per_step': dt / a.steps * 1e3, 'batch_per_gpu': a.batch, 'mode': a.mode, 'fused_loss': bool(tr.fused_loss and tr._fused_ok()), 'dtype': 'f32' if a.gemm == 'fp32' else 'bf16 GEMM operands, f32 accumulate/master',
                          'first_loss': losses[0], 'last_loss': losses[-1], 'cpu_baseline': cpu, 'pipelined': tr.pipelined,
                          'graph_of_last_step': {'nodes': tr.h.n_phar + tr.h.n_pocket, 'edges': tr.h.query('train_edges'),
                                                 'coord_edges': tr.h.query('train_coord_edges')},
                          'allreduce_ms': allreduce, 'phase_ms': {'loss_and_grad': ev[0].elapsed_time(ev[1]), 'allreduce': ev[1].elapsed_time(ev[2]),
                                       'clip_and_adamw': ev[2].elapsed_time(ev[3])}}))


if __name__ == '__main__':
    main()
