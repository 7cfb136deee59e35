"""Pockets shard embarrassingly across the GPUs of a node: every pocket's chain touches only
its own nodes (edges never cross samples, dynamics.py:143), so sampling needs NO data-path
collective.  One process per GPU takes a contiguous block of pockets; device noise is keyed
by the GLOBAL pocket index, so results do not depend on the sharding.  The only
torch.distributed traffic is the optional gather of finished samples to rank 0.

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
    """Contiguous blocks with near-equal summed cost (e.g. (Np+Nl)*degree per pocket)."""
    cost = np.asarray(cost, dtype=np.float64)
    cum = np.concatenate([[0.0], np.cumsum(cost)])
    total = cum[-1]
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        k = int(np.searchsorted(cum, target))
        k = min(max(k, cuts[-1]), len(cost))
        cuts.append(k)
    cuts.append(len(cost))
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


def sample_sharded(sample_fn: Callable, pocket: Dict[str, torch.Tensor], num_nodes_phar,
                   rank: int, world: int, gather: bool = True, group=None, **kw):
    """Run `sample_fn(sub_pocket, sub_num_nodes_phar, pocket_ids=global ids, **kw)` on this
    rank's block and (optionally) gather (xh_phar, xh_pocket) of all ranks in pocket order.

    `sample_fn` is ConditionalDDPM.sample_given_pocket in production."""
    n = len(pocket['size'])
    lo, hi = shard_bounds(n, world)[rank]
    sub, nph = slice_pocket(pocket, num_nodes_phar, lo, hi)
    ids = list(range(lo, hi))
    xh_phar, xh_pocket, phar_mask, pocket_mask = sample_fn(sub, nph, pocket_ids=ids, **kw)
    if not gather or world == 1:
        return xh_phar, xh_pocket, phar_mask + lo, pocket_mask + lo
    import torch.distributed as dist
    outs = [None] * world
    dist.all_gather_object(outs, (xh_phar.cpu(), xh_pocket.cpu(), (phar_mask + lo).cpu(), (pocket_mask + lo).cpu()),
                           group=group)
    return tuple(torch.cat([o[i] for o in outs]) for i in range(4))
