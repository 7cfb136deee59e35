"""Training driver - counterpart of the reference's ``train.py`` without Lightning / W&B.

    python -m cmdgen_amd.train --config configs/crossdocked_ca_cond.yml [--resume <ckpt>]
    python -m torch.distributed.run --nproc-per-node 8 -m cmdgen_amd.train --config ...      # data parallel over RCCL

Same YAML keys as the reference configs (train.py:46-91, configs/*.yml): run_name, logdir, dataset, datadir,
batch_size, lr, n_epochs, gpus, clip_grad, mode, pocket_representation, egnn_params, diffusion_params, eval_epochs,
eval_params (wandb_params / enable_progress_bar / num_sanity_val_steps are accepted and ignored).  Reads
``<datadir>/{train,val}.npz`` and ``size_distribution.npy`` (process_crossdock_ca_only.py:195-207, :355-357).

What Lightning did, done here: epochs over a shuffled loader (each rank takes every world-th batch of the same
permutation), ``HipTrainer.training_step`` (forward, backward, flat-bucket all-reduce, adaptive clipping, AdamW amsgrad),
a validation pass of the eval-mode loss after every epoch (lightning_modules.py:262-287), ``checkpoints/last.ckpt`` and
``best-model-epoch=NN.ckpt`` by validation loss (train.py:93-101) in the Lightning checkpoint format the sampler
loads, metrics as JSON lines in ``<logdir>/<run_name>/metrics.jsonl``.
"""
from __future__ import annotations

import argparse
import json
import os
import time
import warnings
from argparse import Namespace
from pathlib import Path

import numpy as np
import torch
import yaml

from .lightning_modules import PharPocketDDPM
from .training import HipTrainer


def merge_args_and_yaml(args, config_dict):
    """Config values win over command-line ones; nested sections become Namespaces (behaviour of train.py:17-28)."""
    ns = vars(args)
    clash = [k for k in config_dict if k in ns]
    if clash:
        warnings.warn('config file overrides command line for: ' + ', '.join(f'{k}={ns[k]!r}->{config_dict[k]!r}' for k in clash))
    ns.update({k: (Namespace(**v) if isinstance(v, dict) else v) for k, v in config_dict.items()})
    return args


def merge_configs(config, resume_config):
    """Hyper-parameters stored in the checkpoint win over the YAML on --resume (behaviour of train.py:31-40)."""
    stored = {k: (vars(v) if isinstance(v, Namespace) else v) for k, v in resume_config.items()}
    changed = [k for k, v in stored.items() if k in config and config[k] != v]
    if changed:
        warnings.warn('checkpoint hyper-parameters override the config for: ' + ', '.join(changed))
    config.update(stored)
    return config


def _batches(dataset, batch_size, epoch, rank, world, shuffle=True, seed=0, drop_tail=True):
    """every rank walks the same permutation and takes every world-th batch; in training the ragged tail is dropped
    so that all ranks take the same number of optimizer steps (every step is a collective), in validation it is kept
    (ranks may then see different batch counts: only the final sums are reduced)"""
    n = len(dataset)
    order = np.random.Generator(np.random.PCG64(seed + epoch)).permutation(n) if shuffle else np.arange(n)
    starts = list(range(0, n, batch_size))
    usable = (len(starts) // world) * world if (world > 1 and drop_tail) else len(starts)
    for bi in range(rank, usable, world):
        idx = order[starts[bi]:starts[bi] + batch_size]
        yield dataset.collate_fn([dataset[int(i)] for i in idx])


@torch.no_grad()
def validate(model, dataset, batch_size, rank, world):
    """mean eval-mode nll over the validation set (lightning_modules.py:262-287)"""
    model.eval()
    tot, cnt = 0.0, 0
    for batch in _batches(dataset, batch_size, 0, rank, world, shuffle=False, drop_tail=False):
        nll, _ = model.forward(batch)
        tot += float(nll.sum()); cnt += len(nll)
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([tot, cnt], dtype=torch.float64, device=model.device)
        from .collectives import wait_collective
        wait_collective(dist.all_reduce(t, async_op=True))
        tot, cnt = float(t[0]), float(t[1])
    model.train()
    return tot / cnt if cnt > 0 else float('nan')      # an empty validation set must not look like a perfect model


def save_ckpt(model, trainer, path, epoch, best):
    ck = {'state_dict': model.state_dict(), 'hyper_parameters': dict(model.hparams), 'epoch': epoch,
          'hip_trainer': {'exp_avg': trainer.exp_avg.cpu(), 'exp_avg_sq': trainer.exp_avg_sq.cpu(),
                          'max_exp_avg_sq': trainer.max_exp_avg_sq.cpu(), 'step_count': trainer.step_count,
                          'gradnorm_queue': list(trainer.gradnorm_queue.items), 'best_val': best}}
    torch.save(ck, path)


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--config', type=str, required=True)
    p.add_argument('--resume', type=str, default=None)
    p.add_argument('--max_steps', type=int, default=None, help='stop after this many optimizer steps (smoke runs)')
    p.add_argument('--gemm', default='fp32', choices=['fp32', 'bf16'])
    args = p.parse_args(argv)
    with open(args.config, 'r') as f:
        config = yaml.safe_load(f)
    assert 'resume' not in config
    resume = None
    if args.resume is not None:
        resume = torch.load(args.resume, map_location='cpu', weights_only=False)
        config = merge_configs(config, resume['hyper_parameters'])
    gemm, max_steps, resume_path = args.gemm, args.max_steps, args.resume
    args = merge_args_and_yaml(args, config)

    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('nccl')             # RCCL on ROCm
    out_dir = Path(args.logdir, args.run_name)
    histogram = np.load(Path(args.datadir, 'size_distribution.npy')).tolist()
    model = PharPocketDDPM(
        outdir=out_dir, dataset=args.dataset, datadir=args.datadir, batch_size=args.batch_size, lr=args.lr,
        egnn_params=args.egnn_params, diffusion_params=args.diffusion_params, num_workers=args.num_workers,
        augment_noise=args.augment_noise, augment_rotation=args.augment_rotation, clip_grad=args.clip_grad,
        eval_epochs=args.eval_epochs, eval_params=args.eval_params, mode=args.mode, node_histogram=histogram,
        pocket_representation=args.pocket_representation)
    if args.augment_noise > 0 or args.augment_rotation:
        raise NotImplementedError('augment_noise / augment_rotation raise in the reference too (lightning_modules.py:246-254)')
    if resume is not None:
        model.load_state_dict(resume['state_dict'], strict=True)
    model = model.to(torch.device('cuda', local))
    model.setup('fit')
    trainer = HipTrainer(model, gemm_dtype=gemm)
    trainer.pipelined = True            # a step does not wait for its own gradient norm (training.HipTrainer.optimizer_step)
    start_epoch, best = 0, float('inf')
    if resume is not None and 'hip_trainer' in resume:
        st = resume['hip_trainer']
        trainer.exp_avg.copy_(st['exp_avg']); trainer.exp_avg_sq.copy_(st['exp_avg_sq'])
        trainer.max_exp_avg_sq.copy_(st['max_exp_avg_sq'])
        trainer.step_count = int(st['step_count'])
        trainer.gradnorm_queue.items = list(st['gradnorm_queue'])
        start_epoch, best = int(resume.get('epoch', -1)) + 1, float(st.get('best_val', float('inf')))
    trainer.broadcast_state(0)          # replicas start from rank 0's parameters / moments, as under DDP
    if rank == 0:
        (out_dir / 'checkpoints').mkdir(parents=True, exist_ok=True)
        log = open(out_dir / 'metrics.jsonl', 'a')
    torch.manual_seed(1234 + rank)
    done = False
    for epoch in range(start_epoch, args.n_epochs):
        t0 = time.perf_counter()
        losses = []
        for batch in _batches(model.train_dataset, args.batch_size, epoch, rank, world):
            # the collate output is on the host: keep its node counts there, so a step never waits on the device for them
            batch = dict(batch)
            batch['num_phar_atoms_cpu'], batch['num_pocket_nodes_cpu'] = batch['num_phar_atoms'].cpu(), batch['num_pocket_nodes'].cpu()
            info = trainer.training_step(batch)
            losses.append(info['loss'])             # device scalars: read once per epoch, the steps stay pipelined
            if max_steps is not None and trainer.step_count >= max_steps:
                done = True
                break
        losses = [float(x) for x in losses]
        trainer._collect_norm()             # the last step's norm enters the queue before it is checkpointed
        val = validate(model, model.val_dataset, args.batch_size, rank, world)
        sampled = None
        if rank == 0 and (epoch + 1) % int(args.eval_epochs) == 0:
            # validation sampling on rank 0 every eval_epochs (validation_epoch_end, lightning_modules.py:289-304)
            tic = time.perf_counter()
            model.eval()
            sampled = model.sample_and_analyze_given_pocket(int(vars(args.eval_params).get('n_eval_samples', 16)),
                                                            model.val_dataset, batch_size=model.eval_batch_size)
            model.train()
            sampled['evaluation_s'] = time.perf_counter() - tic
            print(f"Evaluation took {sampled['evaluation_s']:.2f} seconds")
        if rank == 0:
            rec = {'epoch': epoch, 'loss/train': float(np.mean(losses)) if losses else None, 'loss/val': val,
                   'steps': trainer.step_count, 'epoch_s': time.perf_counter() - t0}
            if sampled is not None:
                rec.update({f'{k}/val': v for k, v in sampled.items()})
            log.write(json.dumps(rec) + '\n'); log.flush()
            print(rec)
            save_ckpt(model, trainer, out_dir / 'checkpoints' / 'last.ckpt', epoch, min(best, val))
            if val == val and val < best:
                for old in (out_dir / 'checkpoints').glob('best-model-epoch=*.ckpt'):
                    old.unlink()
                save_ckpt(model, trainer, out_dir / 'checkpoints' / f'best-model-epoch={epoch:02d}.ckpt', epoch, val)
        best = min(best, val) if val == val else best
        if done:
            break
    if world > 1:
        from .collectives import host_barrier
        host_barrier()
        dist.destroy_process_group()
    return {'epochs': epoch + 1 - start_epoch, 'steps': trainer.step_count, 'best_val': best, 'out_dir': str(out_dir),
            'resumed_from': resume_path}


if __name__ == '__main__':
    main()
