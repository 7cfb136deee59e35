"""Shared helpers for the parity tests (test infrastructure)."""
import os

import numpy as np
import torch

import cmdgen_amd  # noqa: F401  (alias shim)
from cmdgen_amd.synthetic import ModelConfig, make_state_dict, make_pockets

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name)) as f:
        return {k: f[k] for k in f.files}


def cases_of(g):
    return sorted({k.split('/')[0] for k in g if '/' in k})


def cfg_from_meta(H, L, R, timesteps=500, simple=False):
    return ModelConfig(hidden_nf=int(H), n_layers=int(L), residue_nf=int(R), timesteps=timesteps, no_com_projection=simple)


def masks_from_sizes(pocket_size, num_nodes_phar):
    B = len(pocket_size)
    return (np.repeat(np.arange(B, dtype=np.int64), num_nodes_phar),
            np.repeat(np.arange(B, dtype=np.int64), pocket_size))


def dynamics_case(g, name):
    """-> cfg, numpy state dict, inputs dict for one G2 case."""
    H, L, B, R, seed, gain1, first = [int(v) for v in g[name + '/meta']]
    cfg = cfg_from_meta(H, L, R)
    sd = make_state_dict(cfg, seed=seed, coord_gain=1.0 if gain1 else 1e-3)
    pm, qm = masks_from_sizes(g[name + '/pocket_size'], g[name + '/num_nodes_phar'])
    inp = dict(xh_phar=g[name + '/xh_phar'], xh_pocket=g[name + '/xh_pocket'], t=g[name + '/t'],
               mask_phar=pm, mask_pocket=qm)
    return cfg, sd, inp


def chain_case(g, name):
    H, L, B, R, seed, K, gain1, first = [int(v) for v in g[name + '/meta']]
    cfg = cfg_from_meta(H, L, R, simple=name.startswith('simple'))
    sd = make_state_dict(cfg, seed=seed, coord_gain=1.0 if gain1 else 1e-3)
    rep = 'CA' if R == 20 else 'full-atom'
    ragged = bool(int(g[name + '/ragged']))
    if rep == 'full-atom':
        pb = make_pockets(B, rep, n_pocket_nodes=90, n_phar=9, first_index=first)
    else:
        pb = make_pockets(B, rep, ragged=ragged, n_phar=8, first_index=first)
    return cfg, sd, pb, K


class NoiseTape:
    """Replays recorded Gaussian draws in order (noise-injection interface)."""
    def __init__(self, arr):
        self.arr, self.i = arr, 0

    def __call__(self, shape):
        out = torch.from_numpy(self.arr[self.i].copy())
        assert tuple(out.shape) == tuple(shape)
        self.i += 1
        return out


def rms(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)))


HIST = np.zeros((30, 70), dtype=np.float64)
for _i in range(3, 26):
    for _j in range(20, 66):
        HIST[_i, _j] = 1 + ((_i * 7 + _j * 3) % 11)       # the histogram make_golden.py uses


def loss_case(g):
    H, L, B, R, seed, first = [int(v) for v in g['meta']]
    cfg = cfg_from_meta(H, L, R)
    sd = make_state_dict(cfg, seed=seed, coord_gain=1.0)
    pb = make_pockets(B, 'CA', ragged=True, first_index=first)
    nl = g['num_nodes_phar']
    phar = {'x': torch.from_numpy(g['phar_x'].copy()), 'one_hot': torch.from_numpy(g['phar_one_hot'].copy()),
            'size': torch.from_numpy(nl.copy()), 'mask': torch.from_numpy(np.repeat(np.arange(B), nl))}
    pocket = {'x': torch.from_numpy(pb.x.copy()), 'one_hot': torch.from_numpy(pb.one_hot.copy()),
              'size': torch.from_numpy(pb.size.copy()), 'mask': torch.from_numpy(pb.mask.copy())}
    return cfg, sd, phar, pocket, HIST


# ---------------------------------------------------------------- joint model (G9)
class JointNoiseTape:
    """Replays packed combined draws ([D, Nl*(3+P) + Np*(3+R)], make_golden_joint.pack_draws) as the
    three randn calls the reference makes per combined draw: x [Nl+Np,3], h_phar [Nl,P], h_pocket [Np,R]."""
    def __init__(self, arr, Nl, Np, P=8, R=20):
        self.arr, self.Nl, self.Np, self.P, self.R = arr, Nl, Np, P, R
        self.i = 0          # combined draws consumed
        self.sub = 0

    def __call__(self, shape):
        row = self.arr[self.i]
        a = row[:self.Nl * (3 + self.P)].reshape(self.Nl, 3 + self.P)
        b = row[self.Nl * (3 + self.P):].reshape(self.Np, 3 + self.R)
        out = [np.concatenate([a[:, :3], b[:, :3]]), a[:, 3:], b[:, 3:]][self.sub]
        assert tuple(out.shape) == tuple(shape), (out.shape, shape)
        self.sub += 1
        if self.sub == 3:
            self.sub, self.i = 0, self.i + 1
        return torch.from_numpy(np.ascontiguousarray(out))


def joint_cfg(H, L, R=20, timesteps=500):
    return ModelConfig(hidden_nf=int(H), n_layers=int(L), residue_nf=int(R), timesteps=timesteps,
                       update_pocket_coords=True)


def joint_cases(g, kind):
    return sorted({k.split('/')[1] for k in g if k.startswith(kind + '/')})


def joint_inpaint_case(g, name):
    H, L, B, R, seed, K, resamplings, jump, first = [int(v) for v in g[f'inpaint/{name}/meta']]
    cfg = joint_cfg(H, L, R)
    sd = make_state_dict(cfg, seed=seed, coord_gain=1.0)
    pb = make_pockets(B, 'CA', ragged=True, first_index=first)
    nl = pb.num_nodes_phar
    phar = {'x': g[f'inpaint/{name}/phar_x'], 'one_hot': g[f'inpaint/{name}/phar_one_hot'],
            'size': nl, 'mask': np.repeat(np.arange(B, dtype=np.int64), nl)}
    pocket = {'x': pb.x, 'one_hot': pb.one_hot, 'size': pb.size, 'mask': pb.mask}
    return cfg, sd, phar, pocket, K, resamplings, jump


def joint_loss_case(g):
    H, L, B, R, seed, first = [int(v) for v in g['loss/meta']]
    cfg = joint_cfg(H, L, R)
    sd = make_state_dict(cfg, seed=seed, coord_gain=1.0)
    pb = make_pockets(B, 'CA', ragged=True, first_index=first)
    nl = g['loss/num_nodes_phar']
    phar = {'x': torch.from_numpy(g['loss/phar_x'].copy()), 'one_hot': torch.from_numpy(g['loss/phar_one_hot'].copy()),
            'size': torch.from_numpy(nl.copy()), 'mask': torch.from_numpy(np.repeat(np.arange(B), nl))}
    pocket = {'x': torch.from_numpy(g['loss/pocket_x'].copy()), 'one_hot': torch.from_numpy(pb.one_hot.copy()),
              'size': torch.from_numpy(pb.size.copy()), 'mask': torch.from_numpy(pb.mask.copy())}
    return cfg, sd, phar, pocket, HIST


LOSS_NAMES = ['delta_log_px', 'error_t_phar', 'error_t_pocket', 'SNR_weight', 'loss_0_x_phar', 'loss_0_x_pocket',
              'loss_0_h', 'neg_log_constants', 'kl_prior', 'log_pN', 't_int', 'xh_phar_hat']


# ---------------------------------------------------------------- round-2 goldens (make_golden_r2.py)
def bounded_case(g, name):
    """G13: chains of a model with noise_precision 0.05 / norm_values [1, 0.5] (|x| stays O(10 A))."""
    H, L, B, R, seed, K, gain1, first, T = [int(v) for v in g[name + '/meta']]
    cfg = ModelConfig(hidden_nf=H, n_layers=L, residue_nf=R, timesteps=T, noise_precision=float(g[name + '/noise_precision']),
                      norm_values=tuple(float(v) for v in g[name + '/norm_values']))
    sd = make_state_dict(cfg, seed=seed, coord_gain=1.0 if gain1 else 1e-3)
    pb = make_pockets(B, 'CA', ragged=bool(int(g[name + '/ragged'])), n_phar=8, first_index=first)
    return cfg, sd, pb, K


def fullsize_chain_case(g, name='chain_fa366_K5'):
    """G12: BASELINE configs[4]'s real shape - 366 full-atom pocket atoms + 15 phar points per sample."""
    H, L, B, R, seed, K, gain1, first = [int(v) for v in g[name + '/meta']]
    cfg = cfg_from_meta(H, L, R)
    sd = make_state_dict(cfg, seed=seed, coord_gain=1.0 if gain1 else 1e-3)
    pb = make_pockets(B, 'full-atom', n_phar=15, first_index=first)
    return cfg, sd, pb, K


def g5_case(g):
    H, L, B, R, seed, gain1, first = [int(v) for v in g['meta']]
    cfg = cfg_from_meta(H, L, R)
    sd = make_state_dict(cfg, seed=seed, coord_gain=1.0 if gain1 else 1e-3)
    pm, qm = masks_from_sizes(g['pocket_size'], g['num_nodes_phar'])
    inp = dict(xh_phar=g['xh_phar'], xh_pocket=g['xh_pocket'], t=g['t'], mask_phar=pm, mask_pocket=qm)
    return cfg, sd, inp


def pocket_dict(pb):
    return {'x': torch.from_numpy(pb.x), 'one_hot': torch.from_numpy(pb.one_hot),
            'size': torch.from_numpy(pb.size), 'mask': torch.from_numpy(pb.mask)}
