"""world_size-2 gloo test of the multi-GPU path (CPU): pockets shard with no data-path
collective and the gathered result equals the single-process result.  The per-pocket sampler
is the oracle here (test infrastructure standing in for the GPU), with noise keyed by the
GLOBAL pocket index exactly as the device Philox stream is."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import cmdgen_amd  # noqa: E402,F401
from cmdgen_amd import sharding  # noqa: E402
from cmdgen_amd.synthetic import ModelConfig, make_state_dict, make_pockets  # noqa: E402


def oracle_sampler(cfg, sd, K, seen=None):
    from oracle import ref_cpu
    p = ref_cpu.to_torch_params(sd)

    def fn(pocket, num_nodes_phar, pocket_ids=None, **kw):
        if seen is not None:
            seen.append(kw.get('seed'))           # the Philox seed sample_sharded agreed on (rank 0's, by a tensor broadcast)
        nph = torch.as_tensor(num_nodes_phar)
        # per-pocket noise streams keyed by global id -> independent of the sharding
        draws = []
        for d in range(K + 2):
            parts = [torch.randn((int(n), 11), generator=torch.Generator().manual_seed(1000 * int(g) + d))
                     for g, n in zip(pocket_ids, nph)]
            draws.append(torch.cat(parts))
        it = iter(draws)
        with torch.no_grad():
            return ref_cpu.sample_given_pocket(p, cfg.as_dict(), pocket, nph, timesteps=K,
                                               noise=lambda shape: next(it))
    return fn


def _worker(rank, world, port, q, n_pockets=5):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    cfg = ModelConfig(hidden_nf=64, n_layers=2)
    sd = make_state_dict(cfg, seed=5)
    pb = make_pockets(n_pockets, 'CA', ragged=True)
    pocket = {'x': torch.from_numpy(pb.x), 'one_hot': torch.from_numpy(pb.one_hot),
              'size': torch.from_numpy(pb.size), 'mask': torch.from_numpy(pb.mask)}
    seen = []
    torch.manual_seed(100 + rank)                 # the ranks' global generators differ: each would draw another seed
    out = sharding.sample_sharded(oracle_sampler(cfg, sd, 3, seen), pocket, pb.num_nodes_phar, rank, world)
    seeds = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(seeds, torch.tensor([seen[0] if seen else -1], dtype=torch.int64))
    if rank == 0:
        q.put([o.numpy() for o in out] + [np.array([int(t) for t in seeds])])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_sampling_equals_single_process():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = ModelConfig(hidden_nf=64, n_layers=2)
    sd = make_state_dict(cfg, seed=5)
    pb = make_pockets(5, 'CA', ragged=True)
    pocket = {'x': torch.from_numpy(pb.x), 'one_hot': torch.from_numpy(pb.one_hot),
              'size': torch.from_numpy(pb.size), 'mask': torch.from_numpy(pb.mask)}
    want = sharding.sample_sharded(oracle_sampler(cfg, sd, 3), pocket, pb.num_nodes_phar, 0, 1)
    seeds = got.pop()
    assert seeds[0] == seeds[1] and seeds[0] >= 0    # both ranks sampled with rank 0's seed
    for a, b in zip(got, want):
        assert np.array_equal(a, b.numpy())          # bit-identical: shards are independent


def test_rank_with_an_empty_block_still_joins_the_gather():
    """fewer pockets than ranks: the rank without pockets runs no chain, contributes zero rows of the right width and
    the gathered result is the single-process result (it used to call the sampler with an empty batch and leave the
    other ranks waiting in all_gather)"""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, 1)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = ModelConfig(hidden_nf=64, n_layers=2)
    sd = make_state_dict(cfg, seed=5)
    pb = make_pockets(1, 'CA', ragged=True)
    pocket = {'x': torch.from_numpy(pb.x), 'one_hot': torch.from_numpy(pb.one_hot),
              'size': torch.from_numpy(pb.size), 'mask': torch.from_numpy(pb.mask)}
    want = sharding.sample_sharded(oracle_sampler(cfg, sd, 3), pocket, pb.num_nodes_phar, 0, 1)
    seeds = got.pop()
    assert seeds[0] >= 0 and seeds[1] == -1          # (the rank with the empty block ran no chain)
    for a, b in zip(got, want):
        assert np.array_equal(a, b.numpy())


def test_skewed_costs_leave_no_rank_empty():
    # the searchsorted cuts alone gave [0, 1, 1, 1, 4] here: ranks 1 and 2 without pockets although there are 4 for 4 ranks
    b = sharding.balanced_shard_bounds([100, 1, 1, 1], 4)
    assert b == [(0, 1), (1, 2), (2, 3), (3, 4)]
    # 8 full-atom pockets on 8 GPUs, the first one 2.5x the others
    b = sharding.balanced_shard_bounds([2.5] + [1.0] * 7, 8)
    assert all(hi - lo == 1 for lo, hi in b) and b[0][0] == 0 and b[-1][1] == 8
    rng = np.random.default_rng(0)
    for _ in range(200):
        n, world = int(rng.integers(1, 40)), int(rng.integers(1, 9))
        cost = rng.pareto(0.7, size=n) + 1e-3
        b = sharding.balanced_shard_bounds(cost, world)
        assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(world - 1))
        if n >= world:
            assert all(hi > lo for lo, hi in b)
        else:
            assert all(hi >= lo for lo, hi in b)


def test_shard_bounds():
    assert sharding.shard_bounds(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert sharding.shard_bounds(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    b = sharding.balanced_shard_bounds([1, 1, 1, 1, 4, 4], 2)
    assert b[0][0] == 0 and b[-1][1] == 6 and b[0][1] == b[1][0]
    cost = np.array([1, 1, 1, 1, 4, 4], float)
    assert abs(cost[b[0][0]:b[0][1]].sum() - cost[b[1][0]:b[1][1]].sum()) <= 4
    pb = make_pockets(4, 'CA', ragged=True)
    pocket = {'x': torch.from_numpy(pb.x), 'one_hot': torch.from_numpy(pb.one_hot),
              'size': torch.from_numpy(pb.size), 'mask': torch.from_numpy(pb.mask)}
    sub, nph = sharding.slice_pocket(pocket, pb.num_nodes_phar, 1, 3)
    assert sub['size'].tolist() == pb.size[1:3].tolist() and sub['mask'].min() == 0 and sub['mask'].max() == 1
    assert len(sub['x']) == int(pb.size[1:3].sum()) and nph.tolist() == pb.num_nodes_phar[1:3].tolist()
