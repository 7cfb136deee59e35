"""Golden for the TRAINING step: parameter gradients of the reference's training loss and the
parameters after optimizer steps, produced by running the real reference (ConditionalDDPM.forward in train mode,
autograd, torch.optim.AdamW(amsgrad=True, weight_decay=1e-12) as lightning_modules.py:141-143, norm clipping as
:543-568) in the build container.  Writes tests/golden/g11_train.npz.

    python tests/golden/make_golden_grad.py

The loss assembly (l2 objective, training branch) restates lightning_modules.py:198-215 / :253-254 - ten lines of
arithmetic on the reference's own loss terms; everything else is the reference's code and autograd.
Inputs are the G6 loss case (g6_loss.npz: t_int and the Gaussian draw pinned, t = 0 and t = T included).
"""
import os
import sys

import numpy as np
import torch
import torch._dynamo  # noqa: F401  (must be imported before make_golden installs its sys.modules stubs)
import torch.optim  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import HIST, build_reference_ddpm, import_reference, pockets_to_torch  # noqa: E402

from cmdgen_amd.synthetic import ModelConfig, make_pockets  # noqa: E402


def training_loss(ddpm, phar, pocket):
    terms = ddpm(phar, pocket, return_info=True)
    delta_log_px, error_t_phar, error_t_pocket, SNR_weight, loss_0_x_phar, loss_0_x_pocket, loss_0_h, \
        neg_log_const_0, kl_prior, log_pN, t_int, xh_phar_hat, info = terms
    x_dims = 3
    error_t_phar = error_t_phar / ((x_dims + ddpm.phar_nf) * phar['size'])          # lightning_modules.py:199-203
    error_t_pocket = error_t_pocket / ((x_dims + ddpm.residue_nf) * pocket['size'])
    loss_t = 0.5 * (error_t_phar + error_t_pocket)
    loss_0 = loss_0_x_phar / (x_dims * phar['size']) + loss_0_x_pocket / (x_dims * pocket['size']) + loss_0_h   # :206-208
    nll = loss_t + loss_0 + kl_prior                                                   # :217
    return nll.mean(0), nll                                                            # :254


def main():
    mods = import_reference()
    g6 = np.load(os.path.join(HERE, 'g6_loss.npz'))
    H, L, B, R, seed, first = [int(v) for v in g6['meta']]
    cfg = ModelConfig(hidden_nf=H, n_layers=L, timesteps=500)
    ddpm, _ = build_reference_ddpm(mods, cfg, seed, 1.0, HIST)
    ddpm.train()
    pb = make_pockets(B, 'CA', ragged=True, first_index=first)
    nl = g6['num_nodes_phar']
    pmask = np.repeat(np.arange(B), nl)
    out = {'meta': g6['meta']}
    params = [(n, p) for n, p in ddpm.named_parameters()]
    opt = torch.optim.AdamW([p for _, p in params], lr=1e-3, amsgrad=True, weight_decay=1e-12)
    real_randint = torch.randint
    queue = [3000.0]                                                                   # lightning_modules.py:78-80
    for step in range(3):
        draws = iter([g6['eps0']])
        ddpm.sample_gaussian = lambda size, device: torch.from_numpy(next(draws).copy())
        torch.randint = lambda lo, hi, size, device=None: torch.from_numpy(g6['t_int'].copy())
        phar = {'x': torch.from_numpy(g6['phar_x'].copy()), 'one_hot': torch.from_numpy(g6['phar_one_hot'].copy()),
                'size': torch.from_numpy(nl.copy()), 'mask': torch.from_numpy(pmask.copy())}
        pocket = pockets_to_torch(pb)
        opt.zero_grad()
        loss, nll = training_loss(ddpm, phar, pocket)
        loss.backward()
        torch.randint = real_randint
        out[f'step{step}/loss'] = loss.detach().numpy()
        out[f'step{step}/nll'] = nll.detach().numpy()
        if step == 0:
            for n, p in params:
                out[f'grad/{n}'] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
        # adaptive clipping, lightning_modules.py:543-568 (+ torch.nn.utils.clip_grad_norm_, what Lightning calls)
        max_grad_norm = 1.5 * np.mean(queue) + 2 * np.std(queue)
        grads = [p.grad for _, p in params if p.grad is not None]
        grad_norm = torch.norm(torch.stack([torch.norm(gr.detach(), 2.0) for gr in grads]), 2.0)
        # make the clipping bite at step 1 so that the clipped branch is pinned too
        if step == 1:
            max_grad_norm = 0.5 * float(grad_norm)
        torch.nn.utils.clip_grad_norm_([p for _, p in params], max_grad_norm)
        queue.insert(0, float(max_grad_norm) if float(grad_norm) > max_grad_norm else float(grad_norm))
        out[f'step{step}/grad_norm'] = np.asarray(float(grad_norm))
        out[f'step{step}/max_grad_norm'] = np.asarray(float(max_grad_norm))
        opt.step()
        print('step', step, 'loss', float(loss), 'grad_norm', float(grad_norm), 'max', max_grad_norm)
    for n, p in params:
        out[f'param_after3/{n}'] = p.detach().numpy().copy()
    nz = sum(int(np.abs(out[f'grad/{n}']).max() > 0) for n, _ in params)
    print('parameters', len(params), 'with non-zero grad', nz)
    np.savez_compressed(os.path.join(HERE, 'g11_train.npz'), **out)


if __name__ == '__main__':
    main()
