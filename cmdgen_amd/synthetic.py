"""Deterministic synthetic inputs for the DiffPhar denoising path.

No trained checkpoint and no dataset ship with the reference, so
every test, golden vector and benchmark in this repo uses

* ``make_state_dict``  - a seeded weight set with exactly the tensor names and
  shapes of the reference checkpoint's ``state_dict`` (prefix ``ddpm.``), i.e.
  the parameters created by ``EGNNDynamics.__init__`` (dynamics.py:10-73),
  ``EGNN.__init__`` (egnn_new.py:160-191), ``GCL.__init__`` (egnn_new.py:7-29),
  ``EquivariantUpdate.__init__`` (egnn_new.py:70-85) and the gamma table of
  ``PredefinedNoiseSchedule`` (en_diffusion.py:1157-1184);
* ``make_pockets``     - CrossDocked-shaped synthetic pockets:
  ``Np`` points uniform in a 12 A ball, residue / atom types drawn with the
  dataset frequencies of constants.py:104-107 / :119-122, one numpy PCG64
  stream per *global pocket index* so a shard sees the same pockets whatever
  the sharding.

Only numpy is used so that the GPU box regenerates bit-identical weights from
the seed (fixtures hold inputs/outputs, never weights).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np


@dataclass
class ModelConfig:
    """Hyper-parameters that fix every shape on the hot path.

    Defaults are the shipped C-alpha config, configs/crossdocked_ca_cond.yml:22-41.
    """
    phar_nf: int = 8
    residue_nf: int = 20           # 20 for 'CA', 11 for 'full-atom'
    joint_nf: int = 32
    hidden_nf: int = 256
    n_layers: int = 5
    n_dims: int = 3
    edge_cutoff: Optional[float] = 6.0
    attention: bool = True
    tanh: bool = True
    norm_constant: float = 1.0
    inv_sublayers: int = 1
    sin_embedding: bool = False
    aggregation_method: str = 'sum'
    normalization_factor: float = 100.0
    coords_range: float = 15.0     # egnn_new.py:161 default; quirk Q3: never divided by n_layers
    condition_time: bool = True
    update_pocket_coords: bool = False   # conditional mode (lightning_modules.py:125)
    no_com_projection: bool = False      # True: SimpleConditionalDDPM (conditional_model.py:481-525)
    # diffusion
    timesteps: int = 500
    noise_schedule: str = 'polynomial_2'
    noise_precision: float = 1e-5
    norm_values: Sequence[float] = (1.0, 4.0)
    norm_biases: Sequence[Optional[float]] = (None, 0.0)

    def as_dict(self):
        return dict(self.__dict__)


def linear_specs(cfg: ModelConfig) -> "OrderedDict[str, tuple]":
    """(out_features, in_features, has_bias) for every nn.Linear on the path,
    keyed by the reference's module path below ``ddpm.dynamics.`` and listed in
    the reference's registration order."""
    P, R, J, H = cfg.phar_nf, cfg.residue_nf, cfg.joint_nf, cfg.hidden_nf
    dyn_nf = J + (1 if cfg.condition_time else 0)
    s = OrderedDict()
    s['phar_encoder.0'] = (2 * P, P, True)
    s['phar_encoder.2'] = (J, 2 * P, True)
    s['phar_decoder.0'] = (2 * P, J, True)
    s['phar_decoder.2'] = (P, 2 * P, True)
    s['residue_encoder.0'] = (2 * R, R, True)
    s['residue_encoder.2'] = (J, 2 * R, True)
    s['residue_decoder.0'] = (2 * R, J, True)
    s['residue_decoder.2'] = (R, 2 * R, True)
    s['egnn.embedding'] = (H, dyn_nf, True)
    s['egnn.embedding_out'] = (dyn_nf, H, True)
    edge_in = 2 * H + (24 if cfg.sin_embedding else 2)     # [h_row | h_col | radial | d0], egnn_new.py:35,146; sin_embedding: 12 + 12 features (:174-176)
    for b in range(cfg.n_layers):
        for g in range(cfg.inv_sublayers):
            p = f'egnn.e_block_{b}.gcl_{g}.'
            s[p + 'edge_mlp.0'] = (H, edge_in, True)
            s[p + 'edge_mlp.2'] = (H, H, True)
            s[p + 'node_mlp.0'] = (H, 2 * H, True)
            s[p + 'node_mlp.2'] = (H, H, True)
            if cfg.attention:
                s[p + 'att_mlp.0'] = (1, H, True)
        p = f'egnn.e_block_{b}.gcl_equiv.'
        s[p + 'coord_mlp.0'] = (H, edge_in, True)
        s[p + 'coord_mlp.2'] = (H, H, True)
        s[p + 'coord_mlp.4'] = (1, H, False)
    return s


def gamma_table(noise_schedule: str, timesteps: int, precision: float) -> np.ndarray:
    """fp32 gamma lookup table, built in float64 then cast.

    Restates ``polynomial_schedule`` / ``clip_noise_schedule`` /
    ``PredefinedNoiseSchedule.__init__`` (en_diffusion.py:1135-1149, :1119-1132,
    :1157-1184).  Only the polynomial family is used by the shipped configs.
    """
    if 'polynomial' not in noise_schedule:
        raise ValueError(noise_schedule)
    parts = noise_schedule.split('_')
    assert len(parts) == 2
    power = float(parts[1])
    steps = timesteps + 1
    grid = np.linspace(0, steps, steps)
    a2 = (1.0 - np.power(grid / steps, power)) ** 2
    # ratio clipping (alpha_t / alpha_{t-1} in [1e-3, 1]) then re-accumulate
    a2 = np.concatenate([np.ones(1), a2], axis=0)
    ratio = np.clip(a2[1:] / a2[:-1], a_min=0.001, a_max=1.0)
    a2 = np.cumprod(ratio, axis=0)
    a2 = (1.0 - 2.0 * precision) * a2 + precision
    s2 = 1.0 - a2
    gamma = -(np.log(a2) - np.log(s2))
    return gamma.astype(np.float32)


def make_state_dict(cfg: ModelConfig, seed: int = 0, coord_gain: float = 1e-3,
                    prefix: str = 'ddpm.') -> "OrderedDict[str, np.ndarray]":
    """Seeded weights with the reference checkpoint's names/shapes.

    uniform(+-1/sqrt(fan_in)) for weights and biases (the nn.Linear default
    family); the coordinate head ``coord_mlp.4`` uses the xavier-uniform bound
    ``gain*sqrt(6/(fan_in+fan_out))`` with gain 1e-3 as in egnn_new.py:76-77.
    ``coord_gain=1.0`` gives a trained-like magnitude for coordinate updates
    (used by parity tests so that moved coordinates actually matter).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = OrderedDict()
    sd[prefix + 'buffer'] = np.zeros(1, dtype=np.float32)
    if cfg.noise_schedule == 'learned':
        # GammaNetwork (en_diffusion.py:1058-1074): PositiveLinear(1, 1), (1, 1024), (1024, 1) with the weights offset by -2 before
        # the softplus, gamma_0 / gamma_1; its own stream, so the other tensors do not depend on the schedule
        grng = np.random.Generator(np.random.PCG64(7_000_003 + seed))
        for nm, (fo, fi) in (('l1', (1, 1)), ('l2', (1024, 1)), ('l3', (1, 1024))):
            b = 1.0 / math.sqrt(fi)
            sd[prefix + f'gamma.{nm}.weight'] = (grng.uniform(-b, b, size=(fo, fi)) - 2.0).astype(np.float32)
            sd[prefix + f'gamma.{nm}.bias'] = grng.uniform(-b, b, size=(fo,)).astype(np.float32)
        sd[prefix + 'gamma.gamma_0'] = np.asarray([-5.0], dtype=np.float32)
        sd[prefix + 'gamma.gamma_1'] = np.asarray([10.0], dtype=np.float32)
    else:
        sd[prefix + 'gamma.gamma'] = gamma_table(cfg.noise_schedule, cfg.timesteps,
                                                 cfg.noise_precision)
    for name, (fo, fi, has_bias) in linear_specs(cfg).items():
        if name.endswith('coord_mlp.4'):
            bound = coord_gain * math.sqrt(6.0 / (fi + fo))
        else:
            bound = 1.0 / math.sqrt(fi)
        sd[prefix + 'dynamics.' + name + '.weight'] = rng.uniform(
            -bound, bound, size=(fo, fi)).astype(np.float32)
        if has_bias:
            sd[prefix + 'dynamics.' + name + '.bias'] = rng.uniform(
                -bound, bound, size=(fo,)).astype(np.float32)
    return sd


# dataset type frequencies (constants.py:119-122 'aa_hist', :104-107 'aa_hist' of crossdock_full)
_AA_HIST = [277175, 92406, 254046, 201833, 234995, 376966, 147704, 290683, 173210,
            421883, 157813, 174241, 148581, 120232, 173848, 274430, 247605, 326134,
            88552, 226668]
_ATOM_HIST = [23481798, 6139100, 6753114, 278864, 0, 0, 0, 0, 0, 0, 0]


@dataclass
class PocketBatch:
    """Flat PyG-style batch as the reference builds it (lightning_modules.py:439-455)."""
    x: np.ndarray          # [sum Np, 3] float32
    one_hot: np.ndarray    # [sum Np, R] float32 (0/1, un-normalised)
    size: np.ndarray       # [B] int64
    mask: np.ndarray       # [sum Np] int64 ascending
    num_nodes_phar: np.ndarray  # [B] int64
    pocket_index: np.ndarray = field(default=None)  # [B] global pocket ids


def make_pockets(n_pockets: int, representation: str = 'CA', n_pocket_nodes: int = None,
                 n_phar: int = 15, ragged: bool = False, first_index: int = 0,
                 radius: float = 12.0) -> PocketBatch:
    """Synthetic CrossDocked-shaped pockets, one PCG64 stream per global index."""
    if representation == 'CA':
        R, hist, default_np = 20, _AA_HIST, 44
    elif representation == 'full-atom':
        R, hist, default_np = 11, _ATOM_HIST, 366
    else:
        raise ValueError(representation)
    prob = np.asarray(hist, dtype=np.float64)
    prob = prob / prob.sum()
    xs, hs, sizes, nph = [], [], [], []
    for k in range(n_pockets):
        g = first_index + k
        rng = np.random.Generator(np.random.PCG64(1_000_003 + g))
        if ragged:
            npk = int(rng.integers(30, 61)) if representation == 'CA' else int(rng.integers(250, 451))
            nl = int(rng.integers(5, 26))
        else:
            npk = default_np if n_pocket_nodes is None else int(n_pocket_nodes)
            nl = int(n_phar)
        # uniform in a ball: direction * r, r = R * u^(1/3)
        v = rng.normal(size=(npk, 3))
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        r = radius * np.cbrt(rng.uniform(size=(npk, 1)))
        xs.append((v * r).astype(np.float32))
        t = rng.choice(R, size=npk, p=prob)
        hs.append(np.eye(R, dtype=np.float32)[t])
        sizes.append(npk)
        nph.append(nl)
    size = np.asarray(sizes, dtype=np.int64)
    return PocketBatch(
        x=np.concatenate(xs, axis=0), one_hot=np.concatenate(hs, axis=0), size=size,
        mask=np.repeat(np.arange(n_pockets, dtype=np.int64), size),
        num_nodes_phar=np.asarray(nph, dtype=np.int64),
        pocket_index=np.arange(first_index, first_index + n_pockets, dtype=np.int64))


def make_training_batch(n_complexes: int, first_index: int, representation: str = 'CA', ragged: bool = True) -> Dict[str, np.ndarray]:
    """One synthetic training batch in the NPZ / collate schema of the dataset path (dataset.py:7-64,
    process_crossdock_ca_only.py:195-207): ragged CrossDocked-shaped pockets, pharmacophore points scattered 2.5 A around
    the pocket centre, uniform types.  numpy arrays; keys as PharPocketDDPM.get_phar_and_pocket reads them, plus the node
    counts once more under '<name>_cpu' (what a collate function has on the host before the batch moves to the device)."""
    pb = make_pockets(n_complexes, representation, ragged=ragged, first_index=first_index)
    rng = np.random.Generator(np.random.PCG64(first_index))
    nl = pb.num_nodes_phar
    pm = np.repeat(np.arange(n_complexes), nl)
    com = np.stack([pb.x[pb.mask == b].mean(0) for b in range(n_complexes)])
    px = (com[pm] + rng.normal(size=(len(pm), 3)) * 2.5).astype(np.float32)
    poh = np.eye(8, dtype=np.float32)[rng.integers(0, 8, size=len(pm))]
    return {'phar_coords': px, 'phar_one_hot': poh, 'num_phar_atoms': nl, 'phar_mask': pm,
            'pocket_c_alpha': pb.x, 'pocket_one_hot': pb.one_hot, 'num_pocket_nodes': pb.size, 'pocket_mask': pb.mask,
            'num_phar_atoms_cpu': nl, 'num_pocket_nodes_cpu': pb.size}


def min_cutoff_margin(x: np.ndarray, mask: np.ndarray, cutoff: float) -> float:
    """Smallest | ||x_i - x_j|| - cutoff | over same-sample pairs (float64).

    Parity fixtures require a margin well above fp32 round-off because the
    radius graph is a hard threshold (SURVEY.md §7 'Hard parts' #2)."""
    best = np.inf
    x = x.astype(np.float64)
    for b in np.unique(mask):
        p = x[mask == b]
        d = np.sqrt(((p[:, None, :] - p[None, :, :]) ** 2).sum(-1))
        iu = np.triu_indices(len(p), k=1)
        if len(iu[0]):
            best = min(best, float(np.abs(d[iu] - cutoff).min()))
    return best
