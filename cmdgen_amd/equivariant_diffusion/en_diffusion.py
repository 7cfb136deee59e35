"""Noise schedule, alpha/sigma algebra and node-count prior
(counterparts in en_diffusion.py: PredefinedNoiseSchedule :1152-1188, polynomial_schedule :1135-1149,
clip_noise_schedule :1119-1132, EnVariationalDiffusion helpers :79-103, :849-949,
DistributionNodes :952-1022).

These are scalar, per-chain quantities: they are evaluated on the host with the same torch
fp32 ops as the reference (so the per-step table handed to the HIP library is bit-identical)
and never sit on the per-step critical path.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .. import utils
from ..synthetic import gamma_table


class PredefinedNoiseSchedule(nn.Module):
    """Lookup table gamma[0..T] for the 'polynomial_<p>' schedules (built in float64, stored fp32)."""
    def __init__(self, noise_schedule, timesteps, precision):
        super().__init__()
        self.timesteps = timesteps
        if 'polynomial' not in noise_schedule:
            raise ValueError(noise_schedule)       # 'cosine' / 'learned' are not used by the shipped configs
        self.gamma = nn.Parameter(torch.from_numpy(gamma_table(noise_schedule, timesteps, precision)),
                                  requires_grad=False)

    def forward(self, t):
        return self.gamma[torch.round(t * self.timesteps).long()]


class PositiveLinear(nn.Module):
    """Linear layer whose weight enters through softplus (en_diffusion.py:1025-1056)."""
    def __init__(self, in_features, out_features, weight_init_offset=-2):
        super().__init__()
        bound = 1.0 / math.sqrt(in_features)
        self.weight = nn.Parameter(torch.empty(out_features, in_features).uniform_(-bound, bound) + weight_init_offset)
        self.bias = nn.Parameter(torch.empty(out_features).uniform_(-bound, bound))

    def forward(self, x):
        return F.linear(x, F.softplus(self.weight), self.bias)


class GammaNetwork(nn.Module):
    """noise_schedule='learned' (en_diffusion.py:1058-1096): a monotone network t -> gamma(t), normalised to
    [gamma_0, gamma_1].  A scalar per-chain quantity like the predefined table: evaluated on the host with the reference's
    torch ops; the sampler's per-step scalars (step_table) are built from it and handed to the HIP library."""
    def __init__(self):
        super().__init__()
        self.l1, self.l2, self.l3 = PositiveLinear(1, 1), PositiveLinear(1, 1024), PositiveLinear(1024, 1)
        self.gamma_0 = nn.Parameter(torch.tensor([-5.]))
        self.gamma_1 = nn.Parameter(torch.tensor([10.]))

    def gamma_tilde(self, t):
        l1_t = self.l1(t)
        return l1_t + self.l3(torch.sigmoid(self.l2(l1_t)))

    def forward(self, t):
        zeros, ones = torch.zeros_like(t), torch.ones_like(t)
        g0, g1, gt = self.gamma_tilde(zeros), self.gamma_tilde(ones), self.gamma_tilde(t)
        return self.gamma_0 + (self.gamma_1 - self.gamma_0) * ((gt - g0) / (g1 - g0))

    def table(self, timesteps):
        """gamma(i / T), i = 0..T, fp32 on the host: what the HIP handle stores where a predefined schedule has its lookup table."""
        with torch.no_grad():
            t = (torch.arange(timesteps + 1, dtype=torch.float32) / timesteps).view(-1, 1)
            return self.cpu_copy()(t).view(-1).numpy().astype(np.float32)

    def cpu_copy(self):
        """a host copy for scalar evaluations.  Built under fork_rng: PositiveLinear.__init__ draws ~2k uniform values from the GLOBAL torch
        generator, and the reference's sampler leaves that stream alone - host-side randn after a sampling call must not depend on how often
        this helper ran"""
        with torch.random.fork_rng(devices=[]):
            g = GammaNetwork()
        g.load_state_dict({k: v.detach().cpu() for k, v in self.state_dict().items()})
        return g


class DistributionNodes:
    """Joint histogram over (n_phar, n_pocket) node counts with +1e-3 smoothing (en_diffusion.py:952-1022)."""
    def __init__(self, histogram):
        hist = torch.tensor(np.asarray(histogram)).float() + 1e-3
        prob = hist / hist.sum()
        self.prob = prob
        n1, n2 = prob.shape
        self.idx_to_n_nodes = torch.stack(torch.meshgrid(torch.arange(n1), torch.arange(n2), indexing='ij'),
                                          dim=-1).view(-1, 2)
        self.n_nodes_to_idx = {tuple(x.tolist()): i for i, x in enumerate(self.idx_to_n_nodes)}
        self.m = torch.distributions.Categorical(prob.view(-1), validate_args=True)
        self.n1_given_n2 = [torch.distributions.Categorical(prob[:, j], validate_args=True) for j in range(n2)]
        self.n2_given_n1 = [torch.distributions.Categorical(prob[i, :], validate_args=True) for i in range(n1)]
        # the conditional log-probabilities of every (n1, n2) pair, evaluated once by the same Categorical objects:
        # the per-sample Python loop of the reference (:1010-1022) becomes one device gather (no host syncs per step)
        self._lp_n1_given_n2 = torch.stack([m.log_prob(torch.arange(n1)) for m in self.n1_given_n2], dim=1)
        self._lp_n2_given_n1 = torch.stack([m.log_prob(torch.arange(n2)) for m in self.n2_given_n1], dim=0)
        self._lp_cache = {}

    def _table(self, which, device):
        key = (which, str(device))
        if key not in self._lp_cache:
            self._lp_cache[key] = (self._lp_n1_given_n2 if which == 1 else self._lp_n2_given_n1).to(device)
        return self._lp_cache[key]

    def sample(self, n_samples=1):
        idx = self.m.sample((n_samples,))
        a, b = self.idx_to_n_nodes[idx].T
        return a, b

    def sample_conditional(self, n1=None, n2=None):
        assert (n1 is None) ^ (n2 is None), 'Exactly one input argument must be None'
        m = self.n1_given_n2 if n2 is not None else self.n2_given_n1
        c = n2 if n2 is not None else n1
        return torch.tensor([m[int(i)].sample() for i in c], device=c.device)

    def log_prob(self, batch_n_nodes_1, batch_n_nodes_2):
        assert batch_n_nodes_1.dim() == 1 and batch_n_nodes_2.dim() == 1
        idx = torch.tensor([self.n_nodes_to_idx[(a, b)]
                            for a, b in zip(batch_n_nodes_1.tolist(), batch_n_nodes_2.tolist())])
        return self.m.log_prob(idx).to(batch_n_nodes_1.device)

    def log_prob_n1_given_n2(self, n1, n2):
        assert n1.dim() == 1 and n2.dim() == 1
        return self._table(1, n1.device)[n1.long(), n2.long()]

    def log_prob_n2_given_n1(self, n2, n1):
        assert n1.dim() == 1 and n2.dim() == 1
        return self._table(2, n2.device)[n1.long(), n2.long()]


def fresh_seed() -> int:
    """Philox seed of one sampling call when the caller gives none: drawn from torch's global generator, so every
    call gets new noise (the reference draws fresh torch.randn noise per call) while torch.manual_seed(...) still
    makes a run reproducible."""
    return int(torch.randint(0, 2 ** 63 - 1, (1,), dtype=torch.int64).item())


class EnVariationalDiffusion(nn.Module):
    """The joint model (mode 'joint'): schedule algebra shared by the diffusion variants
    (en_diffusion.py:13-103, :849-906) plus the joint sampler and RePaint inpainting
    (en_diffusion.py:576-831), which run as cmdgen_joint_chain on the device.  Needs dynamics built with
    update_pocket_coords=True (lightning_modules.py:125).  The subclass ConditionalDDPM replaces the
    samplers with the pocket-conditioned one (every shipped config uses that)."""
    use_hip_graph = True

    def __init__(self, dynamics: nn.Module, phar_nf: int, residue_nf: int, n_dims: int,
                 size_histogram: Dict, timesteps: int = 1000, parametrization='eps',
                 noise_schedule='learned', noise_precision=1e-4, loss_type='vlb',
                 norm_values=(1., 1.), norm_biases=(None, 0.)):
        super().__init__()
        assert loss_type in {'vlb', 'l2'}
        assert parametrization == 'eps'
        self.learned_schedule = noise_schedule == 'learned'
        if self.learned_schedule:       # (en_diffusion.py:41-46; sampling only here: the training step needs a predefined schedule)
            assert loss_type == 'vlb', 'A noise schedule can only be learned with a vlb objective.'
        self.loss_type = loss_type
        self.gamma = GammaNetwork() if self.learned_schedule else \
            PredefinedNoiseSchedule(noise_schedule, timesteps=timesteps, precision=noise_precision)
        self.dynamics = dynamics
        self.phar_nf, self.residue_nf, self.n_dims = phar_nf, residue_nf, n_dims
        self.num_classes = phar_nf
        self.T = timesteps
        self.parametrization = parametrization
        self.norm_values, self.norm_biases = norm_values, norm_biases
        self.register_buffer('buffer', torch.zeros(1))
        self.size_distribution = DistributionNodes(size_histogram)
        if not self.learned_schedule:
            self.check_issues_norm_values()
        if hasattr(dynamics, 'attach_diffusion'):
            dynamics.attach_diffusion(timesteps, self.gamma_table_host(), norm_values, norm_biases)

    def check_issues_norm_values(self, num_stdevs=8):
        zeros = torch.zeros((1, 1))
        gamma_0 = self.gamma(zeros)
        sigma_0 = self.sigma(gamma_0, target_tensor=zeros).item()
        norm_value = self.norm_values[1]
        if sigma_0 * num_stdevs > 1. / norm_value:
            raise ValueError(f'Value for normalization value {norm_value} probably too large with sigma_0 '
                             f'{sigma_0:.5f} and 1 / norm_value = {1. / norm_value}')

    # ---- alpha / sigma algebra (all on [B,1]-shaped gammas)
    @staticmethod
    def inflate_batch_array(array, target):
        return array.view((array.size(0),) + (1,) * (len(target.size()) - 1))

    def sigma(self, gamma, target_tensor):
        return self.inflate_batch_array(torch.sqrt(torch.sigmoid(gamma)), target_tensor)

    def alpha(self, gamma, target_tensor):
        return self.inflate_batch_array(torch.sqrt(torch.sigmoid(-gamma)), target_tensor)

    @staticmethod
    def SNR(gamma):
        return torch.exp(-gamma)

    def sigma_and_alpha_t_given_s(self, gamma_t, gamma_s, target_tensor):
        sigma2_t_given_s = self.inflate_batch_array(
            -torch.expm1(F.softplus(gamma_s) - F.softplus(gamma_t)), target_tensor)
        log_alpha2_t_given_s = F.logsigmoid(-gamma_t) - F.logsigmoid(-gamma_s)
        alpha_t_given_s = self.inflate_batch_array(torch.exp(0.5 * log_alpha2_t_given_s), target_tensor)
        return sigma2_t_given_s, torch.sqrt(sigma2_t_given_s), alpha_t_given_s

    def gamma_table_host(self) -> np.ndarray:
        """gamma at t = i / T: the predefined lookup table, or the learned network tabulated with its current weights."""
        return self.gamma.table(self.T) if self.learned_schedule else self.gamma.gamma.detach().cpu().numpy()

    def refresh_learned_schedule(self):
        """A learned schedule's weights may have changed since the handle was built (load_state_dict): re-attach its table."""
        if self.learned_schedule and hasattr(self.dynamics, 'attach_diffusion'):
            tab = self.gamma_table_host()
            if not np.array_equal(tab, getattr(self.dynamics, '_gamma', None)):
                self.dynamics.attach_diffusion(self.T, tab, self.norm_values, self.norm_biases,
                                               no_com_projection=bool(self.dynamics._cfg.get('no_com_projection', False)))

    def step_table(self, timesteps: int) -> np.ndarray:
        """[K+1, 4] per-step scalars of the ancestral sampler, evaluated with the reference's
        own op sequence on a single-sample batch (conditional_model.py:345-366, :429-433;
        final row: sigma_0, alpha_0, SNR(-gamma_0/2), t=0 from :108-131)."""
        table = torch.from_numpy(self.gamma_table_host())          # (learned: the tabulation is the cache key, the network is evaluated at t itself)
        T, K = self.T, timesteps
        cache = self.__dict__.setdefault('_step_tables', {})        # K tiny torch ops x 15 per call otherwise
        hit = cache.get(K)
        if hit is not None and torch.equal(hit[0], table):
            return hit[1]
        z = torch.zeros(1, 1)
        rows = []
        if self.learned_schedule:
            net = self.gamma.cpu_copy()
            look = lambda t: net(t).detach()
        else:
            look = lambda t: table[torch.round(t * T).long()]
        for s in reversed(range(K)):
            s_arr = torch.full((1, 1), fill_value=s) / K
            t_arr = (torch.full((1, 1), fill_value=s) + 1) / K
            g_s, g_t = look(s_arr), look(t_arr)
            s2, s_ts, a_ts = self.sigma_and_alpha_t_given_s(g_t, g_s, z)
            sig_s, sig_t = self.sigma(g_s, z), self.sigma(g_t, z)
            rows.append([a_ts.item(), (s2 / a_ts / sig_t).item(), (s_ts * sig_s / sig_t).item(), t_arr.item()])
        g0 = look(torch.zeros(1, 1))
        rows.append([self.sigma(g0, z).item(), self.alpha(g0, z).item(), self.SNR(-0.5 * g0).item(), 0.0])
        out = np.asarray(rows, dtype=np.float32)
        cache[K] = (table.clone(), out)
        return out

    # ---- normalisation (en_diffusion.py:874-906)
    def normalize(self, phar=None, pocket=None):
        if phar is not None:
            phar['x'] = phar['x'] / self.norm_values[0]
            phar['one_hot'] = (phar['one_hot'].float() - self.norm_biases[1]) / self.norm_values[1]
        if pocket is not None:
            pocket['x'] = pocket['x'] / self.norm_values[0]
            pocket['one_hot'] = (pocket['one_hot'].float() - self.norm_biases[1]) / self.norm_values[1]
        return phar, pocket

    def unnormalize(self, x, h_cat):
        return x * self.norm_values[0], h_cat * self.norm_values[1] + self.norm_biases[1]

    def unnormalize_z(self, z_phar, z_pocket):
        nd = self.n_dims
        xl, hl = self.unnormalize(z_phar[:, :nd], z_phar[:, nd:])
        xp, hp = self.unnormalize(z_pocket[:, :nd], z_pocket[:, nd:])
        return torch.cat([xl, hl], dim=1), torch.cat([xp, hp], dim=1)

    def subspace_dimensionality(self, input_size):
        return (input_size - 1) * self.n_dims

    # ---- loss algebra shared with ConditionalDDPM (en_diffusion.py:153-179, :328-330, :467-473, :834-847, :939-949)
    @staticmethod
    def _seg_sum(x, idx, n):
        return torch.zeros((n,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device).index_add_(0, idx, x)

    def sum_except_batch(self, x, indices, n):
        return self._seg_sum(x.sum(-1), indices, n)

    @staticmethod
    def cdf_standard_gaussian(x):
        return 0.5 * (1. + torch.erf(x / math.sqrt(2)))

    @staticmethod
    def gaussian_KL(q_mu_minus_p_mu_squared, q_sigma, p_sigma, d):
        return d * torch.log(p_sigma / q_sigma) + 0.5 * (d * q_sigma ** 2 + q_mu_minus_p_mu_squared) / \
            (p_sigma ** 2) - 0.5 * d

    @staticmethod
    def sample_gaussian(size, device):
        return torch.randn(size, device=device)

    def log_constants_p_x_given_z0(self, n_nodes, device):
        B = len(n_nodes)
        gamma_0 = self.gamma(torch.zeros((B, 1), device=device))
        log_sigma_x = 0.5 * gamma_0.view(B)
        return self.subspace_dimensionality(n_nodes) * (-log_sigma_x - 0.5 * np.log(2 * np.pi))

    def delta_log_px(self, num_nodes):
        return -self.subspace_dimensionality(num_nodes) * np.log(self.norm_values[0])

    def xh_given_zt_and_epsilon(self, z_t, epsilon, gamma_t, batch_mask):
        alpha_t, sigma_t = self.alpha(gamma_t, z_t), self.sigma(gamma_t, z_t)
        return z_t / alpha_t[batch_mask] - epsilon * sigma_t[batch_mask] / alpha_t[batch_mask]

    # ---- the joint model's loss terms (en_diffusion.py:105-151, :181-257, :298-326, :332-465), VALUES only
    def sample_combined_position_feature_noise(self, phar_mask, pocket_mask, eps=None):
        """x noise for all nodes (COM-projected per sample) + feature noise (en_diffusion.py:555-574, :927-937).
        ``eps`` = (raw phar block [Nl, 3+phar_nf], raw pocket block [Np, 3+residue_nf]) to inject a draw."""
        nd, dev = self.n_dims, phar_mask.device
        nl, B = len(phar_mask), int(max(int(phar_mask.max()), int(pocket_mask.max()))) + 1
        if eps is None:
            zx = self.sample_gaussian((nl + len(pocket_mask), nd), dev)
            zh_l = self.sample_gaussian((nl, self.phar_nf), dev)
            zh_p = self.sample_gaussian((len(pocket_mask), self.residue_nf), dev)
        else:
            zx = torch.cat([eps[0][:, :nd], eps[1][:, :nd]]).to(dev)
            zh_l, zh_p = eps[0][:, nd:].to(dev), eps[1][:, nd:].to(dev)
        comb = torch.cat((phar_mask, pocket_mask))
        cnt = self._seg_sum(torch.ones(len(comb), device=dev), comb, B).clamp(min=1)
        zx = zx - (self._seg_sum(zx, comb, B) / cnt[:, None])[comb]
        return torch.cat([zx[:nl], zh_l], dim=1), torch.cat([zx[nl:], zh_p], dim=1)

    def noised_representation(self, xh_phar, xh_pocket, phar_mask, pocket_mask, gamma_t, eps=None):
        alpha_t, sigma_t = self.alpha(gamma_t, xh_phar), self.sigma(gamma_t, xh_phar)
        eps_phar, eps_pocket = self.sample_combined_position_feature_noise(phar_mask, pocket_mask, eps)
        z_t_phar = alpha_t[phar_mask] * xh_phar + sigma_t[phar_mask] * eps_phar
        z_t_pocket = alpha_t[pocket_mask] * xh_pocket + sigma_t[pocket_mask] * eps_pocket
        return z_t_phar, z_t_pocket, eps_phar, eps_pocket

    def kl_prior_with_pocket(self, xh_phar, xh_pocket, mask_phar, mask_pocket, num_nodes):
        B, nd = len(num_nodes), self.n_dims
        gamma_T = self.gamma(torch.ones((B, 1), device=xh_phar.device))
        alpha_T = self.alpha(gamma_T, xh_phar)
        mu_l, mu_p = alpha_T[mask_phar] * xh_phar, alpha_T[mask_pocket] * xh_pocket
        sigma_T = self.sigma(gamma_T, mu_l).squeeze()
        one = torch.ones_like(sigma_T)
        kl_h = self.gaussian_KL(self.sum_except_batch(mu_l[:, nd:] ** 2, mask_phar, B)
                                + self.sum_except_batch(mu_p[:, nd:] ** 2, mask_pocket, B), sigma_T, one, d=1)
        kl_x = self.gaussian_KL(self.sum_except_batch(mu_l[:, :nd] ** 2, mask_phar, B)
                                + self.sum_except_batch(mu_p[:, :nd] ** 2, mask_pocket, B), sigma_T, one,
                                self.subspace_dimensionality(num_nodes))
        return kl_x + kl_h

    def _log_ph_cat(self, z_h, onehot_norm, mask, sigma_0_cat, B, epsilon=1e-10):
        onehot = onehot_norm * self.norm_values[1] + self.norm_biases[1]
        centered = z_h * self.norm_values[1] + self.norm_biases[1] - 1
        lp = torch.log(self.cdf_standard_gaussian((centered + 0.5) / sigma_0_cat[mask])
                       - self.cdf_standard_gaussian((centered - 0.5) / sigma_0_cat[mask]) + epsilon)
        lp = lp - torch.logsumexp(lp, dim=1, keepdim=True)
        return self.sum_except_batch(lp * onehot, mask, B)

    def log_pxh_given_z0_without_constants(self, phar, z_0_phar, eps_phar, net_out_phar, pocket, z_0_pocket,
                                           eps_pocket, net_out_pocket, gamma_0, epsilon=1e-10):
        nd, B = self.n_dims, len(phar['size'])
        sigma_0_cat = self.sigma(gamma_0, target_tensor=z_0_phar) * self.norm_values[1]
        lpx_l = -0.5 * self.sum_except_batch((eps_phar[:, :nd] - net_out_phar[:, :nd]) ** 2, phar['mask'], B)
        lpx_p = -0.5 * self.sum_except_batch((eps_pocket[:, :nd] - net_out_pocket[:, :nd]) ** 2, pocket['mask'], B)
        lph = self._log_ph_cat(z_0_phar[:, nd:], phar['one_hot'], phar['mask'], sigma_0_cat, B, epsilon) + \
            self._log_ph_cat(z_0_pocket[:, nd:], pocket['one_hot'], pocket['mask'], sigma_0_cat, B, epsilon)
        return lpx_l, lpx_p, lph

    def log_pN(self, N_phar, N_pocket):
        return self.size_distribution.log_prob(N_phar, N_pocket)

    @torch.no_grad()
    def forward(self, phar, pocket, return_info=False, t_int=None, eps=None, _net=None):
        """The joint model's 12 loss terms (+ info) of en_diffusion.py:332-465 as VALUES (no autograd graph: the
        HIP evaluation has no backward pass).  ``t_int`` [B,1] and ``eps`` (list of combined draws, each a pair of
        raw blocks, see sample_combined_position_feature_noise) may be supplied for reproducibility."""
        phar, pocket = dict(phar), dict(pocket)
        phar, pocket = self.normalize(phar, pocket)
        B, nd, dev = len(phar['size']), self.n_dims, phar['x'].device
        n_tot = phar['size'] + pocket['size']
        delta_log_px = self.delta_log_px(n_tot)
        if t_int is None:
            t_int = torch.randint(0 if self.training else 1, self.T + 1, size=(B, 1), device=dev).float()
        t_int = t_int.to(dev).float()
        s_int = t_int - 1
        t_is_zero = (t_int == 0).float()
        s, t = s_int / self.T, t_int / self.T
        gamma_s = self.inflate_batch_array(self.gamma(s), phar['x'])
        gamma_t = self.inflate_batch_array(self.gamma(t), phar['x'])
        xh_phar = torch.cat([phar['x'], phar['one_hot']], dim=1)
        xh_pocket = torch.cat([pocket['x'], pocket['one_hot']], dim=1)
        draws = iter(eps) if eps is not None else None
        nxt = (lambda: next(draws)) if draws is not None else (lambda: None)
        pm, qm = phar['mask'], pocket['mask']
        z_l, z_p, e_l, e_p = self.noised_representation(xh_phar, xh_pocket, pm, qm, gamma_t, nxt())
        net_l, net_p = (_net or self.dynamics)(z_l, z_p, t, pm, qm)
        xh_phar_hat = self.xh_given_zt_and_epsilon(z_l, net_l, gamma_t, pm)
        error_l = self.sum_except_batch((e_l - net_l) ** 2, pm, B)
        error_p = self.sum_except_batch((e_p - net_p) ** 2, qm, B)
        SNR_weight = (1 - self.SNR(gamma_s - gamma_t)).squeeze(1)
        assert error_l.size() == SNR_weight.size()
        self._last_train_ctx = {'eps_t': e_l, 'net_out': net_l, 'eps_t_pocket': e_p, 'net_out_pocket': net_p,
                                't_is_zero': t_is_zero, 'SNR_weight': SNR_weight}
        neg_log_constants = -self.log_constants_p_x_given_z0(n_nodes=n_tot, device=dev)
        kl_prior = self.kl_prior_with_pocket(xh_phar, xh_pocket, pm, qm, n_tot)
        if self.training:
            lpx_l, lpx_p, lph = self.log_pxh_given_z0_without_constants(phar, z_l, e_l, net_l, pocket, z_p, e_p, net_p, gamma_t)
            tz = t_is_zero.squeeze()
            loss_0_x_l, loss_0_x_p, loss_0_h = -lpx_l * tz, -lpx_p * tz, -lph * tz
            error_l, error_p = error_l * (1 - t_is_zero).squeeze(), error_p * (1 - t_is_zero).squeeze()
        else:
            t_zeros = torch.zeros_like(s)
            gamma_0 = self.inflate_batch_array(self.gamma(t_zeros), phar['x'])
            z0_l, z0_p, e0_l, e0_p = self.noised_representation(xh_phar, xh_pocket, pm, qm, gamma_0, nxt())
            n0_l, n0_p = self.dynamics(z0_l, z0_p, t_zeros, pm, qm)
            lpx_l, lpx_p, lph = self.log_pxh_given_z0_without_constants(phar, z0_l, e0_l, n0_l, pocket, z0_p, e0_p, n0_p, gamma_0)
            loss_0_x_l, loss_0_x_p, loss_0_h = -lpx_l, -lpx_p, -lph
        log_pN = self.log_pN(phar['size'], pocket['size'])

        def seg_mean(v, m):
            cnt = self._seg_sum(torch.ones(len(m), device=dev), m, B).clamp(min=1)
            return (self._seg_sum(v, m, B) / cnt).mean()
        info = {'eps_hat_phar_x': seg_mean(net_l[:, :nd].abs().mean(1), pm),
                'eps_hat_phar_h': seg_mean(net_l[:, nd:].abs().mean(1), pm),
                'eps_hat_pocket_x': seg_mean(net_p[:, :nd].abs().mean(1), qm),
                'eps_hat_pocket_h': seg_mean(net_p[:, nd:].abs().mean(1), qm)}
        terms = (delta_log_px, error_l, error_p, SNR_weight, loss_0_x_l, loss_0_x_p, loss_0_h, neg_log_constants,
                 kl_prior, log_pN, t_int.squeeze(), xh_phar_hat)
        return (*terms, info) if return_info else terms

    # ---- joint sampler / RePaint inpainting (en_diffusion.py:576-831)
    def get_repaint_schedule(self, resamplings, jump_length, timesteps):
        """RePaint plan (behaviour of en_diffusion.py:649-670): how many denoising steps run before each jump back.
        Walking up in t in strides of ``jump_length``: every stride but the last is denoised ``resamplings`` times -
        once merged into the preceding run, then ``resamplings - 1`` separate re-runs; the list is returned in
        execution order (from t = T down)."""
        runs, t = [], 0
        while t < timesteps:
            final = t + jump_length >= timesteps
            stride = timesteps - t if final else jump_length
            if runs:
                runs[-1] += stride
            else:
                runs.append(stride)
            if not final:
                runs += [jump_length] * (resamplings - 1)
            t += stride
        return runs[::-1]

    def _joint_handle(self, nph, npk, timesteps=None):
        if self.learned_schedule and timesteps is not None and self.T % timesteps != 0:
            # the joint chain's op table takes gamma from the network's TABULATION on the T-grid (gamma[round(step / K * T)]); the reference
            # evaluates the network at step / K itself (en_diffusion.py:599-606).  The two agree exactly when K divides T; otherwise the chain
            # would use gamma at a rounded time, so it is refused (the conditional sampler evaluates the network at step / K: step_table)
            raise NotImplementedError(f"noise_schedule='learned' with the joint sampler needs timesteps that divide T = {self.T} (got {timesteps})")
        if not getattr(self.dynamics, 'update_pocket_coords', False):
            raise ValueError("the joint sampler needs EGNNDynamics(update_pocket_coords=True) (mode 'joint', "
                             "lightning_modules.py:125)")
        self.refresh_learned_schedule()      # (the joint chain's op table is built from the handle's gamma table)
        h = self.dynamics.hip_handle()
        h.set_layout(nph, npk)
        return h

    def _finish_joint(self, h, run, frame_of_step, return_frames):
        (xh_phar, xh_pocket, z_steps), st = h.run_range_guarded(run, h.chain_status)       # (half engine: a NaN reset is re-run on the bf16 split engine first)
        self.last_chain_status = st
        assert st['max_rel_com_error'] < 1e-2, f"Mean is not zero, relative_error {st['max_rel_com_error']}"
        if st['nan_resets']:
            print('Warning: detected nan, resetting EGNN output to zero.')
        if st['max_cog'] > 5e-2 and return_frames == 1:
            print(f"Warning CoG drift with error {st['max_cog']:.3f}. Projecting the positions down.")
        if return_frames == 1:
            return xh_phar, xh_pocket
        # frames (en_diffusion.py:619-623 / :786-791): slot idx <- unnormalize_z(z after that step); slot 0 <- the result
        out_phar = torch.zeros((return_frames,) + tuple(xh_phar.shape), device=xh_phar.device)
        out_pocket = torch.zeros((return_frames,) + tuple(xh_pocket.shape), device=xh_phar.device)
        nl = xh_phar.shape[0] * xh_phar.shape[1]
        for step, idx in frame_of_step:
            zp = z_steps[step, :nl].view_as(xh_phar)
            zq = z_steps[step, nl:].view_as(xh_pocket)
            out_phar[idx], out_pocket[idx] = self.unnormalize_z(zp, zq)
        out_phar[0], out_pocket[0] = xh_phar, xh_pocket
        return out_phar, out_pocket

    @torch.no_grad()
    def sample(self, n_samples, num_nodes_phar, num_nodes_pocket, return_frames=1, timesteps=None,
               device='cuda', noise=None, seed=None, pocket_ids=None):
        """Draw phar AND pocket nodes from the joint model (en_diffusion.py:576-647).

        Extensions: ``noise`` [n_draws, Nl*(3+phar_nf) + Np*(3+residue_nf)] combined Gaussian draws to inject
        (layout of cmdgen_joint_chain), ``seed`` / ``pocket_ids`` for the on-device Philox draws."""
        timesteps = self.T if timesteps is None else timesteps
        assert 0 < return_frames <= timesteps
        assert timesteps % return_frames == 0
        nph = torch.as_tensor(num_nodes_phar).detach().to('cpu', torch.int64).numpy()
        npk = torch.as_tensor(num_nodes_pocket).detach().to('cpu', torch.int64).numpy()
        assert len(nph) == n_samples and len(npk) == n_samples
        h = self._joint_handle(nph, npk, timesteps)
        dev = next(self.dynamics.parameters()).device
        if noise is not None:
            noise = noise.detach().to(dev, torch.float32).contiguous()
        seed = fresh_seed() if seed is None else seed
        run = lambda: h.joint_chain(timesteps, noise=noise, seed=seed, pocket_ids=pocket_ids,
                                    want_steps=return_frames > 1, use_graph=self.use_hip_graph, device=dev)
        frames = [(timesteps - 1 - s, (s * return_frames) // timesteps) for s in range(timesteps)
                  if (s * return_frames) % timesteps == 0]
        out = self._finish_joint(h, run, frames, return_frames)
        phar_mask = utils.num_nodes_to_batch_mask(n_samples, torch.as_tensor(nph), dev)
        pocket_mask = utils.num_nodes_to_batch_mask(n_samples, torch.as_tensor(npk), dev)
        return out[0], out[1], phar_mask, pocket_mask

    @torch.no_grad()
    def inpaint(self, phar, pocket, phar_fixed, pocket_fixed, resamplings=1, jump_length=1, return_frames=1,
                timesteps=None, noise=None, seed=None, pocket_ids=None):
        """RePaint: sample while fixing parts of the input (en_diffusion.py:672-831).  As in the reference the
        inputs are used raw (no normalize call there).  Extensions: noise / seed / pocket_ids as in sample()."""
        timesteps = self.T if timesteps is None else timesteps
        assert 0 < return_frames <= timesteps
        assert timesteps % return_frames == 0
        assert jump_length == 1 or return_frames == 1, "Chain visualization is only implemented for jump_length=1"
        dev = pocket['x'].device
        nph = phar['size'].detach().to('cpu', torch.int64).numpy()
        npk = pocket['size'].detach().to('cpu', torch.int64).numpy()
        for m in (phar['mask'], pocket['mask']):
            if m.numel() > 1 and bool((m[1:] < m[:-1]).any()):
                raise ValueError('batch masks must be ascending and contiguous')
        h = self._joint_handle(nph, npk, timesteps)
        f32 = lambda t: t.detach().to(dev, torch.float32).contiguous()
        if noise is not None:
            noise = f32(noise)
        seed = fresh_seed() if seed is None else seed
        args = dict(phar=(f32(phar['x']), f32(phar['one_hot'])), pocket=(f32(pocket['x']), f32(pocket['one_hot'])),
                    phar_fixed=f32(phar_fixed).reshape(-1), pocket_fixed=f32(pocket_fixed).reshape(-1))
        run = lambda: h.joint_chain(timesteps, resamplings=resamplings, jump_length=jump_length, noise=noise, seed=seed, pocket_ids=pocket_ids,
                                    want_steps=return_frames > 1, use_graph=self.use_hip_graph, **args)
        frames = []
        if return_frames > 1:       # walk the schedule as :723-813 do, noting which steps write a frame
            schedule = self.get_repaint_schedule(resamplings, jump_length, timesteps)
            s, step = timesteps - 1, 0
            for i, n_denoise_steps in enumerate(schedule):
                for j in range(n_denoise_steps):
                    if (n_denoise_steps > jump_length or i == len(schedule) - 1) and (s * return_frames) % timesteps == 0:
                        frames.append((step, (s * return_frames) // timesteps))
                    if j == n_denoise_steps - 1 and i < len(schedule) - 1:
                        s = s + jump_length
                    s -= 1
                    step += 1
        out = self._finish_joint(h, run, frames, return_frames)
        return out[0], out[1], phar['mask'], pocket['mask']
