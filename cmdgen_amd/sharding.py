"""Pockets shard embarrassingly across the GPUs of a node: every pocket's chain touches only
its own nodes (edges never cross samples, dynamics.py:143), so sampling needs NO data-path
collective.  One process per GPU takes a contiguous block of pockets; device noise is keyed
by the GLOBAL pocket index, so results do not depend on the sharding.  The only
torch.distributed traffic is the optional all_gather of the finished samples.

The reference has no multi-GPU sampling code; its DDP gradient
all-reduce belongs to the training row (section 8f #1).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Sequence, Tuple

import numpy as np
import torch


def shard_bounds(n_items: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous blocks, sizes differing by at most one (earlier ranks take the remainder)."""
    base, rem = divmod(n_items, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


def balanced_shard_bounds(cost: Sequence[float], world: int) -> List[Tuple[int, int]]:
    """Contiguous blocks with near-equal summed cost (e.g. (Np+Nl)*degree per pocket).  With at least as many items as
    ranks every rank gets at least one item, however skewed the costs are (an idle GPU helps nobody, and a rank without
    pockets has no chain to run); blocks can be empty only when there are fewer items than ranks, as in `shard_bounds`."""
    cost = np.asarray(cost, dtype=np.float64)
    n = len(cost)
    cum = np.concatenate([[0.0], np.cumsum(cost)])
    total = cum[-1]
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        k = int(np.searchsorted(cum, target))
        if n >= world:
            k = min(max(k, cuts[-1] + 1), n - (world - r))      # leave one item for this rank and one for each later rank
        else:
            k = min(max(k, cuts[-1]), n)
        cuts.append(k)
    cuts.append(n)
    return [(cuts[i], cuts[i + 1]) for i in range(world)]


def slice_pocket(pocket: Dict[str, torch.Tensor], num_nodes_phar, lo: int, hi: int):
    """Sub-batch [lo, hi) of a flat pocket dict (x, one_hot, size, mask) + its phar counts."""
    size = pocket['size']
    starts = torch.cumsum(size, 0) - size
    a = int(starts[lo]) if lo < len(size) else int(size.sum())
    b = int(starts[hi - 1] + size[hi - 1]) if hi > lo else a
    sub = {'x': pocket['x'][a:b], 'one_hot': pocket['one_hot'][a:b], 'size': size[lo:hi],
           'mask': pocket['mask'][a:b] - lo}
    return sub, torch.as_tensor(num_nodes_phar)[lo:hi]


def pocket_cost(num_pocket, num_phar, edge_cutoff=6.0) -> np.ndarray:
    """Relative cost of one pocket's chain: (Np + Nl) * degree, i.e. ~ its edge count.
    The degree estimate is the one cmdgen_set_layout uses to size its grids: C-alpha pockets have ~9 neighbours
    within 6 A, full-atom ones ~36; without a cutoff every sample is a complete graph."""
    n = np.asarray(num_pocket, dtype=np.float64) + np.asarray(num_phar, dtype=np.float64)
    deg = n if edge_cutoff is None else np.minimum(n, np.where(n <= 128, 9.0, 36.0))
    return n * deg


def _gather_rows(t: torch.Tensor, world: int, group=None) -> torch.Tensor:
    """all_gather of per-rank tensors with different row counts (rows concatenated in rank order): one small
    all_gather of the counts, one of the row-padded payload - device tensors stay on the device (RCCL over xGMI
    with the nccl backend, gloo on CPU), nothing is pickled through the host."""
    import torch.distributed as dist
    from .collectives import wait_collective
    # (rows, columns) of every rank: a rank whose block is empty (fewer pockets than ranks) does not know the row width -
    # it takes it from the ranks that have rows
    n = torch.tensor([t.shape[0], t.shape[1] if t.dim() > 1 else 0], dtype=torch.int64, device=t.device)
    shapes = [torch.zeros_like(n) for _ in range(world)]
    wait_collective(dist.all_gather(shapes, n, group=group, async_op=True))
    counts = [int(c[0].item()) for c in shapes]
    if t.shape[0] == 0 and t.dim() > 1:
        t = torch.zeros((0, max(int(c[1].item()) for c in shapes)), dtype=t.dtype, device=t.device)
    cap = max(max(counts), 1)
    pad = torch.zeros((cap,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[:t.shape[0]] = t
    parts = [torch.empty_like(pad) for _ in range(world)]
    wait_collective(dist.all_gather(parts, pad, group=group, async_op=True))
    return torch.cat([p[:c] for p, c in zip(parts, counts)])


def sample_sharded(sample_fn: Callable, pocket: Dict[str, torch.Tensor], num_nodes_phar,
                   rank: int, world: int, gather: bool = True, group=None, balance: bool = True,
                   edge_cutoff=6.0, **kw):
    """Run `sample_fn(sub_pocket, sub_num_nodes_phar, pocket_ids=global ids, **kw)` on this
    rank's block and (optionally) gather (xh_phar, xh_pocket) of all ranks in pocket order.

    Blocks are contiguous and balanced by the pockets' estimated edge counts (`pocket_cost`), not by their number:
    a ragged batch would otherwise leave ranks idle behind the one holding the big pockets.
    `sample_fn` is ConditionalDDPM.sample_given_pocket in production."""
    n = len(pocket['size'])
    if balance:
        cost = pocket_cost(pocket['size'].detach().cpu().numpy(), torch.as_tensor(num_nodes_phar).cpu().numpy(), edge_cutoff)
        lo, hi = balanced_shard_bounds(cost, world)[rank]
    else:
        lo, hi = shard_bounds(n, world)[rank]
    if kw.get('seed') is None and world > 1:
        # device noise is keyed by (seed, global pocket id): the ranks must agree on the seed, and each rank's own global
        # generator need not be in the same state - rank 0 draws it, everybody takes rank 0's
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            from .equivariant_diffusion.en_diffusion import fresh_seed
            from .collectives import wait_collective
            # (a tensor broadcast issued asynchronously, not broadcast_object_list: a blocking collective must not leave its completion event on
            # the stream the chain below is captured on - collectives.wait_collective)
            box = torch.tensor([fresh_seed() if rank == 0 else 0], dtype=torch.int64, device=pocket['size'].device)
            wait_collective(dist.broadcast(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group, async_op=True))
            kw['seed'] = int(box.item())
    sub, nph = slice_pocket(pocket, num_nodes_phar, lo, hi)
    ids = list(range(lo, hi))
    if hi > lo:
        xh_phar, xh_pocket, phar_mask, pocket_mask = sample_fn(sub, nph, pocket_ids=ids, **kw)
    else:
        # an empty block (fewer pockets than ranks): nothing to sample; this rank still takes part in the gathers
        dev = pocket['x'].device
        xh_phar = torch.zeros((0, 0), dtype=torch.float32, device=dev)
        xh_pocket = torch.zeros((0, 3 + pocket['one_hot'].shape[1]), dtype=torch.float32, device=dev)
        phar_mask = torch.zeros((0,), dtype=torch.int64, device=dev)
        pocket_mask = torch.zeros((0,), dtype=torch.int64, device=dev)
    if not gather or world == 1:
        return xh_phar, xh_pocket, phar_mask + lo, pocket_mask + lo
    return (_gather_rows(xh_phar, world, group), _gather_rows(xh_pocket, world, group),
            _gather_rows(phar_mask + lo, world, group), _gather_rows(pocket_mask + lo, world, group))
