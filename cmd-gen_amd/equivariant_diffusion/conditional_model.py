"""ConditionalDDPM: pocket fixed, only pharmacophore nodes diffuse
(counterpart of conditional_model.py:12-475).

``sample_given_pocket`` is the hot path: the whole ancestral chain (init noise, K posterior
steps with one network evaluation each, final decode, drift fix) runs inside
libcmdgen_hip.so with the step captured as a hipGraph; the host only prepares the per-step
scalar table and reads the deferred checks afterwards.
"""
from __future__ import annotations

import numpy as np
import torch

from .en_diffusion import EnVariationalDiffusion
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

    def forward(self, phar, pocket, return_info=False):
        raise NotImplementedError('training loss (conditional_model.py:198-320) is the next scope row '
                                  '(SURVEY.md section 8f #1); round 1 builds the sampling path')

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
          seed       Philox seed for on-device draws (default: torch.initial_seed());
          pocket_ids global pocket indices so a shard draws the same noise as the full batch.
        Returns (xh_phar, xh_pocket, phar_mask, pocket_mask) like the reference; with
        return_frames > 1 the first two carry a leading frame axis.
        """
        timesteps = self.T if timesteps is None else timesteps
        assert 0 < return_frames <= timesteps
        assert timesteps % return_frames == 0
        n_samples = len(pocket['size'])
        device = pocket['x'].device
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
            seed = torch.initial_seed()
        want_steps = return_frames > 1
        xh_phar, xh_pocket, z_steps = h.sample_chain(px, poh, timesteps, noise=noise, seed=seed,
                                                     pocket_ids=pocket_ids, want_steps=want_steps,
                                                     use_graph=self.use_hip_graph)
        # deferred, non-syncing versions of the reference's per-step checks
        st = h.chain_status()
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
        for s in range(timesteps):
            if (s * return_frames) % timesteps == 0:
                idx = (s * return_frames) // timesteps
                zs = z_steps[timesteps - 1 - s]
                nd = self.n_dims
                out_phar[idx] = torch.cat([zs[:, :nd] * self.norm_values[0],
                                           zs[:, nd:] * self.norm_values[1] + self.norm_biases[1]], dim=1)
        out_phar[0], out_pocket[0] = xh_phar, xh_pocket
        return out_phar, out_pocket, phar_mask, pocket['mask']


class SimpleConditionalDDPM(ConditionalDDPM):
    """Variant without the COM subspace trick (conditional_model.py:481-525): not used by the
    shipped configs; the HIP sampler implements the subspace version only."""
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("mode 'pocket_conditioning_simple' is not built (no shipped config uses it)")
