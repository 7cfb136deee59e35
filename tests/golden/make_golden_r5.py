"""G19 (round 5): parameter gradients of the reference's training loss for the EGNN options the training step gained this round -
inv_sublayers > 1 (several GCLs per block, egnn_new.py:127-131, :152-154) and aggregation_method = 'mean' (egnn_new.py:285-292) -
produced by the real reference's forward + autograd in the build container, exactly as make_golden_grad.py produces G11 (same inputs:
the G6 loss case; same loss assembly).  Writes tests/golden/g19_train_options.npz: per case the loss, the per-sample nll and the
gradient of every tensor at step 0.

    python tests/golden/make_golden_r5.py
"""
import os
import sys

import numpy as np
import torch
import torch._dynamo  # noqa: F401  (must be imported before make_golden installs its sys.modules stubs)

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import HIST, build_reference_ddpm, import_reference, pockets_to_torch  # noqa: E402
from make_golden_grad import training_loss  # noqa: E402

from cmdgen_amd.synthetic import ModelConfig, make_pockets  # noqa: E402

CASES = [('s2_sum', 2, 'sum'), ('s1_mean', 1, 'mean'), ('s3_mean', 3, 'mean')]


def main():
    mods = import_reference()
    g6 = np.load(os.path.join(HERE, 'g6_loss.npz'))
    H, L, B, R, seed, first = [int(v) for v in g6['meta']]
    pb = make_pockets(B, 'CA', ragged=True, first_index=first)
    nl = g6['num_nodes_phar']
    pmask = np.repeat(np.arange(B), nl)
    out = {'meta': g6['meta'], 'cases': np.array([c[0] for c in CASES])}
    real_randint = torch.randint
    for name, S, agg in CASES:
        cfg = ModelConfig(hidden_nf=H, n_layers=L, timesteps=500, inv_sublayers=S, aggregation_method=agg)
        ddpm, _ = build_reference_ddpm(mods, cfg, seed, 1.0, HIST)
        ddpm.train()
        params = [(n, p) for n, p in ddpm.named_parameters()]
        draws = iter([g6['eps0']])
        ddpm.sample_gaussian = lambda size, device: torch.from_numpy(next(draws).copy())
        torch.randint = lambda lo, hi, size, device=None: torch.from_numpy(g6['t_int'].copy())
        phar = {'x': torch.from_numpy(g6['phar_x'].copy()), 'one_hot': torch.from_numpy(g6['phar_one_hot'].copy()),
                'size': torch.from_numpy(nl.copy()), 'mask': torch.from_numpy(pmask.copy())}
        pocket = pockets_to_torch(pb)
        loss, nll = training_loss(ddpm, phar, pocket)
        loss.backward()
        torch.randint = real_randint
        out[f'{name}/options'] = np.array([S, int(agg == 'mean')])
        out[f'{name}/loss'] = loss.detach().numpy()
        out[f'{name}/nll'] = nll.detach().numpy()
        nz = 0
        for n, p in params:
            gr = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
            out[f'{name}/grad/{n}'] = gr
            nz += int(np.abs(gr).max() > 0)
        print(name, 'loss', float(loss), 'tensors', len(params), 'with non-zero grad', nz)
    np.savez_compressed(os.path.join(HERE, 'g19_train_options.npz'), **out)


if __name__ == '__main__':
    main()
