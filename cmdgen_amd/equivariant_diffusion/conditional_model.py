"""ConditionalDDPM: pocket fixed, only pharmacophore nodes diffuse
(counterpart of conditional_model.py:12-475).

``sample_given_pocket`` is the hot path: the whole ancestral chain (init noise, K posterior
steps with one network evaluation each, final decode, drift fix) runs inside
libcmdgen_hip.so with the step captured as a hipGraph; the host only prepares the per-step
scalar table and reads the deferred checks afterwards.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from .en_diffusion import EnVariationalDiffusion, fresh_seed
from .. import utils


class ConditionalDDPM(EnVariationalDiffusion):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        assert not self.dynamics.update_pocket_coords
        self.use_hip_graph = True
        self.last_chain_status = None

    # ---- reference entry points that were already stubs there
    def sample_normal(self, *args):
        raise NotImplementedError('Has been replaced by sample_normal_zero_com()')

    def sample_combined_position_feature_noise(self, *args):
        raise NotImplementedError('Use sample_normal_zero_com() instead.')

    def sample(self, *args):
        raise NotImplementedError('Conditional model does not support sampling without given pocket.')

    # ---- loss terms (conditional_model.py:20-106, :158-320) as VALUES: the network evaluation runs in the HIP
    # library and no autograd graph is built here.  Training uses training.HipTrainer: the activation-saving forward
    # (``_net``) plus the library's own backward pass with the analytic gradient of these terms.
    def noised_representation(self, xh_phar, xh0_pocket, phar_mask, pocket_mask, gamma_t, eps=None):
        alpha_t, sigma_t = self.alpha(gamma_t, xh_phar), self.sigma(gamma_t, xh_phar)
        eps_phar = self.sample_gaussian((len(phar_mask), self.n_dims + self.phar_nf), phar_mask.device) \
            if eps is None else eps
        z_t = alpha_t[phar_mask] * xh_phar + sigma_t[phar_mask] * eps_phar
        nd = self.n_dims
        zx, px = self.remove_mean_batch(z_t[:, :nd], xh0_pocket[:, :nd], phar_mask, pocket_mask)
        return torch.cat([zx, z_t[:, nd:]], 1), torch.cat([px, xh0_pocket[:, nd:]], 1), eps_phar

    def kl_prior(self, xh_phar, mask_phar, num_nodes):
        B, nd = len(num_nodes), self.n_dims
        ones = torch.ones((B, 1), device=xh_phar.device)
        gamma_T = self.gamma(ones)
        mu_T = self.alpha(gamma_T, xh_phar)[mask_phar] * xh_phar
        sigma_T = self.sigma(gamma_T, mu_T).squeeze()
        one = torch.ones_like(sigma_T)
        kl_h = self.gaussian_KL(self.sum_except_batch(mu_T[:, nd:] ** 2, mask_phar, B), sigma_T, one, d=1)
        kl_x = self.gaussian_KL(self.sum_except_batch(mu_T[:, :nd] ** 2, mask_phar, B), sigma_T, one,
                                self.subspace_dimensionality(num_nodes))
        return kl_x + kl_h

    def log_pxh_given_z0_without_constants(self, phar, z_0_phar, eps_phar, net_out_phar, gamma_0, epsilon=1e-10):
        nd, B = self.n_dims, len(phar['size'])
        sigma_0_cat = self.sigma(gamma_0, target_tensor=z_0_phar) * self.norm_values[1]
        log_px = -0.5 * self.sum_except_batch((eps_phar[:, :nd] - net_out_phar[:, :nd]) ** 2, phar['mask'], B)
        phar_onehot = phar['one_hot'] * self.norm_values[1] + self.norm_biases[1]
        centered = z_0_phar[:, nd:] * self.norm_values[1] + self.norm_biases[1] - 1
        log_ph = torch.log(self.cdf_standard_gaussian((centered + 0.5) / sigma_0_cat[phar['mask']])
                           - self.cdf_standard_gaussian((centered - 0.5) / sigma_0_cat[phar['mask']]) + epsilon)
        log_ph = log_ph - torch.logsumexp(log_ph, dim=1, keepdim=True)
        return log_px, self.sum_except_batch(log_ph * phar_onehot, phar['mask'], B)

    def log_pN(self, N_phar, N_pocket):
        return self.size_distribution.log_prob_n1_given_n2(N_phar, N_pocket)

    @torch.no_grad()
    def forward(self, phar, pocket, return_info=False, t_int=None, eps=None, _net=None):
        """The 12 loss terms (+ info) of conditional_model.py:198-320 as VALUES (no autograd graph).

        t_int [B,1] and eps (list of the Gaussian draws, one per noised_representation call) may be
        supplied for reproducibility; otherwise they are drawn like the reference does.  ``_net`` replaces the
        network evaluation (training.HipTrainer passes the activation-saving training forward); the tensors the
        analytic loss gradient needs are left in ``self._last_train_ctx``."""
        phar, pocket = dict(phar), dict(pocket)
        phar, pocket = self.normalize(phar, pocket)
        B, nd, dev = len(phar['size']), self.n_dims, phar['x'].device
        delta_log_px = self.delta_log_px(phar['size'])
        if t_int is None:
            t_int = torch.randint(0 if self.training else 1, self.T + 1, size=(B, 1), device=dev).float()
        t_int = t_int.to(dev).float()
        s_int = t_int - 1
        t_is_zero = (t_int == 0).float()
        s, t = s_int / self.T, t_int / self.T
        gamma_s = self.inflate_batch_array(self.gamma(s), phar['x'])
        gamma_t = self.inflate_batch_array(self.gamma(t), phar['x'])
        xh0_phar = torch.cat([phar['x'], phar['one_hot']], dim=1)
        xh0_pocket = torch.cat([pocket['x'], pocket['one_hot']], dim=1)
        cx, cp = self.remove_mean_batch(xh0_phar[:, :nd], xh0_pocket[:, :nd], phar['mask'], pocket['mask'])
        xh0_phar = torch.cat([cx, xh0_phar[:, nd:]], 1)
        xh0_pocket = torch.cat([cp, xh0_pocket[:, nd:]], 1)
        draws = iter(eps) if eps is not None else None
        nxt = (lambda: next(draws).to(dev)) if draws is not None else (lambda: None)
        z_t, xh_pocket, eps_t = self.noised_representation(xh0_phar, xh0_pocket, phar['mask'], pocket['mask'],
                                                           gamma_t, nxt())
        net_out, _ = (_net or self.dynamics)(z_t, xh_pocket, t, phar['mask'], pocket['mask'])
        xh_phar_hat = self.xh_given_zt_and_epsilon(z_t, net_out, gamma_t, phar['mask'])
        error_t = self.sum_except_batch((eps_t - net_out) ** 2, phar['mask'], B)
        SNR_weight = (1 - self.SNR(gamma_s - gamma_t)).squeeze(1)
        assert error_t.size() == SNR_weight.size()
        self._last_train_ctx = {'eps_t': eps_t, 'net_out': net_out, 't_is_zero': t_is_zero, 'SNR_weight': SNR_weight}
        neg_log_constants = -self.log_constants_p_x_given_z0(n_nodes=phar['size'], device=dev)
        kl_prior = self.kl_prior(xh0_phar, phar['mask'], phar['size'])
        if self.training:
            lpx, lph = self.log_pxh_given_z0_without_constants(phar, z_t, eps_t, net_out, gamma_t)
            loss_0_x, loss_0_h = -lpx * t_is_zero.squeeze(), -lph * t_is_zero.squeeze()
            error_t = error_t * (1 - t_is_zero).squeeze()
        else:
            t_zeros = torch.zeros_like(s)
            gamma_0 = self.inflate_batch_array(self.gamma(t_zeros), phar['x'])
            z_0, xh_pocket0, eps_0 = self.noised_representation(xh0_phar, xh0_pocket, phar['mask'], pocket['mask'],
                                                                gamma_0, nxt())
            net_out_0, _ = self.dynamics(z_0, xh_pocket0, t_zeros, phar['mask'], pocket['mask'])
            lpx, lph = self.log_pxh_given_z0_without_constants(phar, z_0, eps_0, net_out_0, gamma_0)
            loss_0_x, loss_0_h = -lpx, -lph
        log_pN = self.log_pN(phar['size'], pocket['size'])
        cnt = self._seg_sum(torch.ones(len(phar['mask']), device=dev), phar['mask'], B).clamp(min=1)
        info = {'eps_hat_phar_x': (self._seg_sum(net_out[:, :nd].abs().mean(1), phar['mask'], B) / cnt).mean(),
                'eps_hat_phar_h': (self._seg_sum(net_out[:, nd:].abs().mean(1), phar['mask'], B) / cnt).mean()}
        terms = (delta_log_px, error_t, torch.tensor(0.0), SNR_weight, loss_0_x, torch.tensor(0.0), loss_0_h,
                 neg_log_constants, kl_prior, log_pN, t_int.squeeze(), xh_phar_hat)
        return (*terms, info) if return_info else terms

    @classmethod
    def remove_mean_batch(cls, x_phar, x_pocket, phar_indices, pocket_indices):
        """Subtract the phar centre of mass from both node sets (conditional_model.py:467-475)."""
        n = int(phar_indices.max()) + 1 if phar_indices.numel() else 0
        tot = torch.zeros((n, x_phar.size(1)), dtype=x_phar.dtype, device=x_phar.device).index_add_(0, phar_indices, x_phar)
        cnt = torch.zeros(n, dtype=x_phar.dtype, device=x_phar.device).index_add_(
            0, phar_indices, torch.ones(len(phar_indices), dtype=x_phar.dtype, device=x_phar.device)).clamp(min=1)
        mean = tot / cnt[:, None]
        return x_phar - mean[phar_indices], x_pocket - mean[pocket_indices]

    @torch.no_grad()
    def sample_given_pocket(self, pocket, num_nodes_phar, return_frames=1, timesteps=None,
                            noise=None, seed=None, pocket_ids=None):
        """Draw samples given pockets (conditional_model.py:388-465).

        Reference arguments: pocket dict(x, one_hot, size, mask), num_nodes_phar [B],
        return_frames, timesteps.  Extensions (keyword-only in spirit):
          noise      [K+2, Nl, 3+phar_nf] Gaussian draws to inject (parity / reproducibility);
          seed       Philox seed for on-device draws (default: a fresh one per call from torch's global generator);
          pocket_ids global pocket indices so a shard draws the same noise as the full batch.
        Returns (xh_phar, xh_pocket, phar_mask, pocket_mask) like the reference; with
        return_frames > 1 the first two carry a leading frame axis.
        """
        timesteps = self.T if timesteps is None else timesteps
        assert 0 < return_frames <= timesteps
        assert timesteps % return_frames == 0
        n_samples = len(pocket['size'])
        device = pocket['x'].device
        self.refresh_learned_schedule()
        h = self.dynamics.hip_handle()
        sizes = pocket['size'].detach().to('cpu', torch.int64).numpy()
        nph = torch.as_tensor(num_nodes_phar).detach().to('cpu', torch.int64).numpy()
        assert len(nph) == n_samples
        pm = pocket['mask']
        if pm.numel() > 1 and bool((pm[1:] < pm[:-1]).any()):
            raise ValueError('pocket mask must be ascending and contiguous')
        h.set_layout(nph, sizes)
        h.set_step_table(timesteps, self.step_table(timesteps))
        phar_mask = utils.num_nodes_to_batch_mask(n_samples, torch.as_tensor(nph), device)
        px = pocket['x'].detach().to(torch.float32).contiguous()
        poh = pocket['one_hot'].detach().to(torch.float32).contiguous()
        if noise is not None:
            noise = noise.detach().to(device, torch.float32).contiguous()
        if seed is None:
            seed = fresh_seed()
        want_steps = return_frames > 1
        # (a NaN reset on the half matrix engine is re-run on the bf16 split engine before it is believed: hip_backend.run_range_guarded)
        (xh_phar, xh_pocket, z_steps), st = h.run_range_guarded(
            lambda: h.sample_chain(px, poh, timesteps, noise=noise, seed=seed, pocket_ids=pocket_ids, want_steps=want_steps,
                                   use_graph=self.use_hip_graph),
            h.chain_status)
        # deferred, non-syncing versions of the reference's per-step checks
        self.last_chain_status = st
        assert st['max_rel_com_error'] < 1e-2, f"Mean is not zero, relative_error {st['max_rel_com_error']}"
        if st['nan_resets']:
            print('Warning: detected nan, resetting EGNN output to zero.')
        if st['max_cog'] > 5e-2 and return_frames == 1:
            print(f"Warning CoG drift with error {st['max_cog']:.3f}. Projecting the positions down.")
        if return_frames == 1:
            return xh_phar, xh_pocket, phar_mask, pocket['mask']
        # frames: idx = s*return_frames//timesteps for steps with (s*return_frames) % timesteps == 0
        # (conditional_model.py:439-442); frame 0 is overwritten by the final sample (:460-461).
        out_phar = torch.zeros((return_frames,) + tuple(xh_phar.shape), device=device)
        out_pocket = torch.zeros((return_frames,) + tuple(xh_pocket.shape), device=device)
        p_steps = h.last_pocket_steps
        nd = self.n_dims
        for s in range(timesteps):
            if (s * return_frames) % timesteps == 0:
                idx = (s * return_frames) // timesteps
                zs, ps = z_steps[timesteps - 1 - s], p_steps[timesteps - 1 - s]          # state after the step with index s
                out_phar[idx] = torch.cat([zs[:, :nd] * self.norm_values[0],
                                           zs[:, nd:] * self.norm_values[1] + self.norm_biases[1]], dim=1)
                out_pocket[idx] = torch.cat([ps * self.norm_values[0], xh_pocket[:, nd:]], dim=1)   # unnormalize_z :897-906
        out_phar[0], out_pocket[0] = xh_phar, xh_pocket
        return out_phar, out_pocket, phar_mask, pocket['mask']


class SimpleConditionalDDPM(ConditionalDDPM):
    """The same model without the subspace trick (conditional_model.py:481-525): the context (pocket) is
    centred once and samples are not projected to the COM-free subspace; translation equivariance comes from
    evaluating everything in the pocket-centred frame."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.dynamics.attach_diffusion(self.T, self.gamma_table_host(), self.norm_values,
                                       self.norm_biases, no_com_projection=True)

    def subspace_dimensionality(self, input_size):
        return input_size * self.n_dims

    @classmethod
    def remove_mean_batch(cls, x_phar, x_pocket, phar_indices, pocket_indices):
        return x_phar, x_pocket

    @staticmethod
    def _pocket_com(pocket):
        n = len(pocket['size'])
        x, m = pocket['x'], pocket['mask']
        tot = torch.zeros((n, x.size(1)), dtype=x.dtype, device=x.device).index_add_(0, m, x)
        cnt = torch.zeros(n, dtype=x.dtype, device=x.device).index_add_(
            0, m, torch.ones(len(m), dtype=x.dtype, device=x.device)).clamp(min=1)
        return tot / cnt[:, None]

    @torch.no_grad()
    def forward(self, phar, pocket, return_info=False, t_int=None, eps=None, _net=None):
        phar, pocket = dict(phar), dict(pocket)
        com = self._pocket_com(pocket)
        phar['x'] = phar['x'] - com[phar['mask']]
        pocket['x'] = pocket['x'] - com[pocket['mask']]
        return super().forward(phar, pocket, return_info, t_int=t_int, eps=eps, _net=_net)

    # sample_given_pocket: the library centres the pocket itself when no_com_projection is set
