#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REAL reference.

Runs only in the build container (needs /root/reference, which never travels
to the GPU box).  The reference is imported read-only with ``sys.modules`` stubs
for the packages that are absent here (torch_scatter, pytorch_lightning, wandb,
rdkit, Bio, openbabel, imageio) - the two torch_scatter functions the path uses
(scatter_add / scatter_mean along dim 0) are restated with ``index_add_``.

Fixtures contain inputs and outputs only; weights are regenerated from a seed
by ``cmdgen_amd.synthetic.make_state_dict`` and loaded into the reference with
``load_state_dict`` (so a fixture never holds reference source or weights).

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz
"""
import importlib
import os
import sys
import types
from argparse import Namespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference/DiffPhar'
sys.path.insert(0, ROOT)

import cmdgen_amd  # noqa: E402
from cmdgen_amd.synthetic import ModelConfig, make_state_dict, make_pockets, min_cutoff_margin  # noqa: E402


# ------------------------------------------------------------------ stubs
class _Anything(types.ModuleType):
    """Attribute-recursive stand-in for an absent package (never computes)."""
    def __getattr__(self, name):
        if name.startswith('__') and name.endswith('__'):
            raise AttributeError(name)
        m = _Anything(self.__name__ + '.' + name)
        setattr(self, name, m)
        return m

    def __call__(self, *a, **k):
        return _Anything(self.__name__ + '()')

    def __iter__(self):
        return iter(())

    def __getitem__(self, k):
        return _Anything(self.__name__ + '[]')

    def __mro_entries__(self, bases):
        return (object,)


def _install_stubs():
    def scatter_add(src, index, dim=0, dim_size=None):
        assert dim == 0
        n = int(index.max()) + 1 if dim_size is None else dim_size
        out = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        return out.index_add_(0, index, src)

    def scatter_mean(src, index, dim=0, dim_size=None):
        assert dim == 0
        n = int(index.max()) + 1 if dim_size is None else dim_size
        tot = scatter_add(src, index, 0, n)
        cnt = torch.zeros(n, dtype=src.dtype, device=src.device).index_add_(
            0, index, torch.ones(len(index), dtype=src.dtype, device=src.device)).clamp(min=1)
        return tot / cnt.view((-1,) + (1,) * (src.dim() - 1))

    ts = types.ModuleType('torch_scatter')
    ts.scatter_add, ts.scatter_mean = scatter_add, scatter_mean
    sys.modules['torch_scatter'] = ts

    pl = types.ModuleType('pytorch_lightning')

    class LightningModule(torch.nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device('cpu')
    pl.LightningModule = LightningModule
    sys.modules['pytorch_lightning'] = pl
    for name in ['wandb', 'rdkit', 'rdkit.Chem', 'rdkit.Chem.rdchem', 'rdkit.Chem.rdMolAlign',
                 'Bio', 'Bio.PDB', 'Bio.PDB.Polypeptide', 'openbabel', 'imageio',
                 'networkx', 'networkx.algorithms', 'seaborn', 'matplotlib',
                 'matplotlib.pyplot', 'rdkit.Chem.Descriptors', 'rdkit.Chem.Crippen',
                 'rdkit.Chem.QED', 'rdkit.DataStructs', 'rdkit.Chem.AllChem',
                 'rdkit.Chem.rdForceFieldHelpers', 'rdkit.Chem.rdMolDescriptors',
                 'rdkit.Chem.rdmolops', 'rdkit.Geometry', 'scipy.ndimage',
                 'rdkit.Chem.rdDetermineBonds', 'mpl_toolkits', 'mpl_toolkits.mplot3d']:
        if name not in sys.modules or name.startswith(('rdkit', 'Bio', 'wandb', 'openbabel', 'imageio', 'seaborn')):
            sys.modules[name] = _Anything(name)


def import_reference():
    _install_stubs()
    sys.path.insert(0, REF)
    mods = {}
    for name in ['equivariant_diffusion.egnn_new', 'equivariant_diffusion.en_diffusion',
                 'equivariant_diffusion.dynamics', 'equivariant_diffusion.conditional_model']:
        mods[name.split('.')[-1]] = importlib.import_module(name)
    return mods


# ------------------------------------------------------------------ builders
def build_reference_ddpm(mods, cfg: ModelConfig, seed, coord_gain, histogram, simple=False, loss_type='l2'):
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):      # reference prints the tables
        dyn = mods['dynamics'].EGNNDynamics(
            phar_nf=cfg.phar_nf, residue_nf=cfg.residue_nf, n_dims=3, joint_nf=cfg.joint_nf,
            hidden_nf=cfg.hidden_nf, device='cpu', act_fn=torch.nn.SiLU(), n_layers=cfg.n_layers,
            attention=cfg.attention, tanh=cfg.tanh, norm_constant=cfg.norm_constant,
            inv_sublayers=cfg.inv_sublayers, sin_embedding=cfg.sin_embedding,
            normalization_factor=cfg.normalization_factor,
            aggregation_method=cfg.aggregation_method, edge_cutoff=cfg.edge_cutoff,
            update_pocket_coords=False)
        cls = mods['conditional_model'].SimpleConditionalDDPM if simple else mods['conditional_model'].ConditionalDDPM
        ddpm = cls(
            dynamics=dyn, phar_nf=cfg.phar_nf, residue_nf=cfg.residue_nf, n_dims=3,
            timesteps=cfg.timesteps, noise_schedule=cfg.noise_schedule,
            noise_precision=cfg.noise_precision, loss_type=loss_type,
            norm_values=list(cfg.norm_values), size_histogram=histogram)
    sd = make_state_dict(cfg, seed=seed, coord_gain=coord_gain, prefix='')
    missing = ddpm.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    ddpm.eval()
    return ddpm, sd


def pockets_to_torch(pb):
    return {'x': torch.from_numpy(pb.x.copy()), 'one_hot': torch.from_numpy(pb.one_hot.copy()),
            'size': torch.from_numpy(pb.size.copy()), 'mask': torch.from_numpy(pb.mask.copy())}


def make_inputs(pb, cfg, rng, phar_radius=5.0):
    """z_phar around each pocket's centre (the geometry a trained model holds)."""
    B = len(pb.size)
    nl = pb.num_nodes_phar
    phar_mask = np.repeat(np.arange(B), nl)
    com = np.stack([pb.x[pb.mask == b].mean(0) for b in range(B)])
    v = rng.normal(size=(len(phar_mask), 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    r = phar_radius * np.cbrt(rng.uniform(size=(len(phar_mask), 1)))
    x = (com[phar_mask] + v * r).astype(np.float32)
    h = rng.normal(size=(len(phar_mask), cfg.phar_nf)).astype(np.float32)
    xh_phar = np.concatenate([x, h], axis=1)
    xh_pocket = np.concatenate([pb.x, pb.one_hot / cfg.norm_values[1]], axis=1).astype(np.float32)
    return xh_phar, xh_pocket, phar_mask.astype(np.int64)


HIST = np.zeros((30, 70), dtype=np.float64)
for _i in range(3, 26):
    for _j in range(20, 66):
        HIST[_i, _j] = 1 + ((_i * 7 + _j * 3) % 11)


def main():
    mods = import_reference()
    en = mods['en_diffusion']
    out = {}

    # ---------------- G1: schedule tables + per-step coefficients
    g1 = {}
    import contextlib, io
    for T in (100, 500, 1000):
        with contextlib.redirect_stdout(io.StringIO()):
            sched = en.PredefinedNoiseSchedule('polynomial_2', timesteps=T, precision=1e-5)
        g1[f'gamma_T{T}'] = sched.gamma.detach().numpy()
    cfg = ModelConfig(hidden_nf=32, n_layers=2, timesteps=500)
    ddpm, _ = build_reference_ddpm(mods, cfg, 0, 1e-3, HIST)
    for K in (5, 50, 500):
        rows = []
        for s in reversed(range(K)):
            s_arr = torch.full((1, 1), fill_value=s) / K
            t_arr = (torch.full((1, 1), fill_value=s) + 1) / K
            g_s, g_t = ddpm.gamma(s_arr), ddpm.gamma(t_arr)
            z = torch.zeros(1, 11)
            s2, s_ts, a_ts = ddpm.sigma_and_alpha_t_given_s(g_t, g_s, z)
            sig_s, sig_t = ddpm.sigma(g_s, z), ddpm.sigma(g_t, z)
            rows.append([a_ts.item(), (s2 / a_ts / sig_t).item(), (s_ts * sig_s / sig_t).item(), t_arr.item()])
        g1[f'coef_T500_K{K}'] = np.asarray(rows, dtype=np.float32)
    g0 = ddpm.gamma(torch.zeros(1, 1))
    g1['final_T500'] = np.asarray([ddpm.sigma(g0, g0).item(), ddpm.alpha(g0, g0).item(),
                                   ddpm.SNR(-0.5 * g0).item()], dtype=np.float32)
    np.savez_compressed(os.path.join(HERE, 'g1_schedule.npz'), **g1)

    # ---------------- G2 / G5: dynamics forward (+ per-block intermediates)
    g2 = {}
    cases = [
        # name, representation, hidden, layers, B, ragged, coord_gain, seed
        ('ca_h32_b3', 'CA', 32, 2, 3, True, 1.0, 11),
        ('ca_h256_b3', 'CA', 256, 5, 3, True, 1.0, 12),
        ('ca_h256_b1', 'CA', 256, 5, 1, False, 1e-3, 13),
        ('ca_h256_b8', 'CA', 256, 5, 8, True, 1.0, 14),
        ('fa_h256_b2', 'full-atom', 256, 5, 2, False, 1.0, 15),
        ('fa_h32_b1', 'full-atom', 32, 3, 1, False, 1.0, 16),
    ]
    for name, rep, H, L, B, ragged, gain, seed in cases:
        cfg = ModelConfig(hidden_nf=H, n_layers=L, residue_nf=20 if rep == 'CA' else 11, timesteps=500)
        ddpm, _ = build_reference_ddpm(mods, cfg, seed, gain, HIST)
        # retry pocket sets until no pair sits within 2e-3 A of the cutoff
        first = 100 * seed
        while True:
            if rep == 'full-atom':
                pb = make_pockets(B, rep, n_pocket_nodes=90, n_phar=9, first_index=first)
            else:
                pb = make_pockets(B, rep, ragged=ragged, n_phar=8, first_index=first)
            rng = np.random.Generator(np.random.PCG64(seed))
            xh_phar, xh_pocket, phar_mask = make_inputs(pb, cfg, rng)
            allx = np.concatenate([xh_phar[:, :3], xh_pocket[:, :3]])
            allm = np.concatenate([phar_mask, pb.mask])
            if min_cutoff_margin(allx, allm, 6.0) > 2e-3:
                break
            first += 1000
        t = rng.uniform(0.05, 0.95, size=(B, 1)).astype(np.float32)
        trace = {}
        # capture intermediates through forward hooks on the reference modules
        egnn = ddpm.dynamics.egnn
        hooks = []
        blocks_h, blocks_x = [], []
        for b in range(L):
            blk = egnn._modules[f'e_block_{b}']
            def _rec(m, i, o, bh=blocks_h, bx=blocks_x):
                bh.append(o[0].detach().numpy().copy())
                bx.append(o[1].detach().numpy().copy())
            hooks.append(blk.register_forward_hook(_rec))
        edges_seen = []
        orig_get_edges = ddpm.dynamics.get_edges

        def rec_edges(mask, x):
            e = orig_get_edges(mask, x)
            edges_seen.append(e.numpy().copy())
            return e
        ddpm.dynamics.get_edges = rec_edges
        with torch.no_grad():
            eps_phar, eps_pocket = ddpm.dynamics(
                torch.from_numpy(xh_phar), torch.from_numpy(xh_pocket), torch.from_numpy(t),
                torch.from_numpy(phar_mask), torch.from_numpy(pb.mask))
        for hk in hooks:
            hk.remove()
        ddpm.dynamics.get_edges = orig_get_edges
        g2[name + '/meta'] = np.asarray([H, L, B, cfg.residue_nf, seed, int(gain == 1.0), first], dtype=np.int64)
        g2[name + '/pocket_size'] = pb.size
        g2[name + '/num_nodes_phar'] = pb.num_nodes_phar
        g2[name + '/xh_phar'] = xh_phar
        g2[name + '/xh_pocket'] = xh_pocket
        g2[name + '/t'] = t
        g2[name + '/eps_phar'] = eps_phar.numpy()
        g2[name + '/edges'] = edges_seen[0].astype(np.int32)
        if H == 32 or name == 'ca_h256_b3':
            g2[name + '/eps_pocket'] = eps_pocket.numpy()
            nl = len(phar_mask)
            for b in range(L):
                # phar rows of h (all 256 cols would be large) + all x rows
                g2[name + f'/block{b}_h_phar'] = blocks_h[b][:nl]
                g2[name + f'/block{b}_h_pocket_head'] = blocks_h[b][nl:nl + 16]
                g2[name + f'/block{b}_x_phar'] = blocks_x[b][:nl]
        print(name, 'E =', edges_seen[0].shape[1], 'max|eps_x| =', float(eps_phar[:, :3].abs().max()))
    np.savez_compressed(os.path.join(HERE, 'g2_dynamics.npz'), **g2)

    # ---------------- G3: get_edges boundary behaviour
    cfg = ModelConfig(hidden_nf=32, n_layers=1)
    ddpm, _ = build_reference_ddpm(mods, cfg, 0, 1e-3, HIST)
    x = np.zeros((40, 3), dtype=np.float32)
    x[1] = [6.0, 0, 0]            # exactly at the cutoff from node 0 (kept: <=)
    x[2] = [0, 6.5, 0]            # outside
    x[3] = [3.0, 4.0, 0]          # 5.0 from node 0; 3-4-5 triangle exact
    rng = np.random.Generator(np.random.PCG64(5))
    x[4:] = rng.uniform(-9, 9, size=(36, 3)).astype(np.float32)
    mask = np.concatenate([np.zeros(30, np.int64), np.ones(10, np.int64)])
    # reorder like the dynamics does: phar rows of all samples first - here simply two samples
    e = ddpm.dynamics.get_edges(torch.from_numpy(mask), torch.from_numpy(x)).numpy()
    np.savez_compressed(os.path.join(HERE, 'g3_edges.npz'), x=x, mask=mask, edges=e.astype(np.int32),
                        margin=np.asarray(min_cutoff_margin(x[4:], mask[4:], 6.0)))

    # ---------------- G4: sampling chains with recorded noise
    g4 = {}
    chain_cases = [
        ('ca_h32_K5', 'CA', 32, 2, 3, 5, 1e-3, 21, True),
        ('ca_h256_K5', 'CA', 256, 5, 4, 5, 1e-3, 22, True),
        ('ca_h256_K50', 'CA', 256, 5, 2, 50, 1e-3, 23, False),
        ('ca_h256_K5_gain1', 'CA', 256, 5, 3, 5, 1.0, 24, True),
        ('fa_h256_K5', 'full-atom', 256, 5, 2, 5, 1e-3, 25, False),
        ('simple_h64_K5', 'CA', 64, 2, 3, 5, 1.0, 26, True),      # SimpleConditionalDDPM (no COM projection)
    ]
    for name, rep, H, L, B, K, gain, seed, ragged in chain_cases:
        cfg = ModelConfig(hidden_nf=H, n_layers=L, residue_nf=20 if rep == 'CA' else 11, timesteps=500)
        ddpm, _ = build_reference_ddpm(mods, cfg, seed, gain, HIST, simple=name.startswith('simple'))
        first, nseed = 100 * seed, seed
        while True:
            if rep == 'full-atom':
                pb = make_pockets(B, rep, n_pocket_nodes=90, n_phar=9, first_index=first)
            else:
                pb = make_pockets(B, rep, ragged=ragged, n_phar=8, first_index=first)
            noises, zs, margins = [], [], []
            gen = torch.Generator().manual_seed(nseed)

            def rec_gauss(size, device):
                n = torch.randn(size, generator=gen)
                noises.append(n.numpy().copy())
                return n
            ddpm.sample_gaussian = rec_gauss
            orig = type(ddpm).sample_p_zs_given_zt.__get__(ddpm)

            def rec_step(s, t, z, xp, pm, qm, fix_noise=False):
                o = orig(s, t, z, xp, pm, qm, fix_noise)
                zs.append(o[0].numpy().copy())
                return o
            ddpm.sample_p_zs_given_zt = rec_step
            orig_edges = type(ddpm.dynamics).get_edges.__get__(ddpm.dynamics)

            def rec_edges(mask, x):     # every network evaluation, incl. the final p(x|z0) one
                margins.append(min_cutoff_margin(x.numpy(), mask.numpy(), 6.0))
                return orig_edges(mask, x)
            ddpm.dynamics.get_edges = rec_edges
            pocket = pockets_to_torch(pb)
            with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
                xh_phar, xh_pocket, phar_mask, pocket_mask = ddpm.sample_given_pocket(
                    pocket, torch.from_numpy(pb.num_nodes_phar), timesteps=K)
            assert len(margins) == K + 1
            if min(margins) > 2e-3:
                break
            first += 1000
            nseed += 1000
        g4[name + '/meta'] = np.asarray([H, L, B, cfg.residue_nf, seed, K, int(gain == 1.0), first], dtype=np.int64)
        g4[name + '/ragged'] = np.asarray(int(ragged))
        g4[name + '/noise'] = np.stack(noises)            # [K+2, Nl, 11]
        g4[name + '/xh_phar'] = xh_phar.numpy()
        g4[name + '/xh_pocket'] = xh_pocket.numpy()
        g4[name + '/phar_mask'] = phar_mask.numpy()
        g4[name + '/min_margin'] = np.asarray(min(margins))
        if K == 5:
            g4[name + '/z_steps'] = np.stack(zs)          # [K, Nl, 11]
        print(name, 'noise draws', len(noises), 'min cutoff margin', min(margins))
    np.savez_compressed(os.path.join(HERE, 'g4_chains.npz'), **g4)

    # ---------------- G6: loss terms of ConditionalDDPM.forward (+ PharPocketDDPM.forward nll), t_int and eps pinned
    for g6_first in range(3100, 99999, 1000):
        g6 = {}
        cfg = ModelConfig(hidden_nf=64, n_layers=2, timesteps=500)
        ddpm, _ = build_reference_ddpm(mods, cfg, 31, 1.0, HIST)
        pb = make_pockets(4, 'CA', ragged=True, first_index=g6_first)
        rng = np.random.Generator(np.random.PCG64(g6_first))
        B = 4
        nl = np.asarray([6, 9, 5, 12], dtype=np.int64)
        phar_mask = np.repeat(np.arange(B), nl)
        com = np.stack([pb.x[pb.mask == b].mean(0) for b in range(B)])
        phar_x = (com[phar_mask] + rng.normal(size=(len(phar_mask), 3)) * 2.0).astype(np.float32)
        phar_oh = np.eye(8, dtype=np.float32)[rng.integers(0, 8, size=len(phar_mask))]
        t_pin = np.asarray([[0.], [137.], [500.], [42.]], dtype=np.float32)       # includes t = 0 and t = T
        eps_pin = [rng.normal(size=(len(phar_mask), 11)).astype(np.float32) for _ in range(2)]
        real_randint = torch.randint
        for mode in ('train', 'eval'):
            ddpm.train() if mode == 'train' else ddpm.eval()
            draws = iter(eps_pin)
            ddpm.sample_gaussian = lambda size, device: torch.from_numpy(next(draws).copy())
            torch.randint = lambda lo, hi, size, device=None: torch.from_numpy(t_pin.copy())
            margins = []
            orig_edges = type(ddpm.dynamics).get_edges.__get__(ddpm.dynamics)

            def rec_edges(mask, x):
                margins.append(min_cutoff_margin(x.numpy(), mask.numpy(), 6.0))
                return orig_edges(mask, x)
            ddpm.dynamics.get_edges = rec_edges
            phar = {'x': torch.from_numpy(phar_x.copy()), 'one_hot': torch.from_numpy(phar_oh.copy()),
                    'size': torch.from_numpy(nl.copy()), 'mask': torch.from_numpy(phar_mask.copy())}
            pocket = pockets_to_torch(pb)
            with torch.no_grad():
                terms = ddpm(phar, pocket, return_info=True)
            torch.randint = real_randint
            names = ['delta_log_px', 'error_t_phar', 'error_t_pocket', 'SNR_weight', 'loss_0_x_phar', 'loss_0_x_pocket',
                     'loss_0_h', 'neg_log_constants', 'kl_prior', 'log_pN', 't_int', 'xh_phar_hat']
            for n, v in zip(names, terms[:-1]):
                g6[f'{mode}/{n}'] = np.asarray(v.numpy() if torch.is_tensor(v) else v, dtype=np.float32)
            g6[f'{mode}/info_eps_hat_phar_x'] = terms[-1]['eps_hat_phar_x'].numpy()
            g6[f'{mode}/info_eps_hat_phar_h'] = terms[-1]['eps_hat_phar_h'].numpy()
            g6[f'{mode}/min_margin'] = np.asarray(min(margins))
            print('G6', mode, 'evaluations', len(margins), 'min margin', min(margins))
        g6['phar_x'], g6['phar_one_hot'], g6['num_nodes_phar'] = phar_x, phar_oh, nl
        g6['t_int'], g6['eps0'], g6['eps1'] = t_pin, eps_pin[0], eps_pin[1]
        g6['meta'] = np.asarray([64, 2, 4, 20, 31, g6_first], dtype=np.int64)
        if min(float(g6['train/min_margin']), float(g6['eval/min_margin'])) > 2e-3:
            np.savez_compressed(os.path.join(HERE, 'g6_loss.npz'), **g6)
            break

    # ---------------- G8: node-count prior
    dn = None
    with contextlib.redirect_stdout(io.StringIO()):
        dn = en.DistributionNodes(HIST)
    n1 = torch.tensor([5, 8, 15, 25, 3])
    n2 = torch.tensor([44, 30, 60, 65, 20])
    lp = dn.log_prob_n1_given_n2(n1, n2).numpy()
    np.savez_compressed(os.path.join(HERE, 'g8_nodes.npz'), hist=HIST, n1=n1.numpy(), n2=n2.numpy(), logp=lp)
    print('done')


if __name__ == '__main__':
    main()
