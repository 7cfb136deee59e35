"""Size-independent properties of the HIP path at BASELINE.json's FULL sizes (configs[1]: 64 C-alpha pockets, the north-star batch of 256,
configs[4]: 256 full-atom pockets of 366 atoms = 97k nodes, ~3.5M edges), where the CPU oracle cannot be run in a test's time:

* E(3) equivariance of EGNNDynamics.forward (dynamics.py:75-139, egnn_new.py:141-157): a REFLECTION x -> -x of every input coordinate
  leaves every squared distance - hence the radius graph, every invariant feature and every summation order - bit-identical, so the
  velocity's x component must come back negated and everything else unchanged, to the run-to-run noise of the float atomics of
  tile-boundary segments; a general rotation + translation must rotate the velocity (tolerance test: the rounding of
  d^2 changes, and a pair within an ulp of the cutoff may flip - a hard threshold the reference has too - so a small share of rows may move);
* batch composition: a pocket's result does not depend on which other pockets share its batch, nor on its position in it (edges never
  cross samples, dynamics.py:141-147; the tile / chunk boundaries of every kernel fall elsewhere for the permuted batch);
* chains: with the Philox stream keyed by the global pocket id, a pocket's sampled pharmacophore is the same whether it is sampled in a
  batch of 256 or in four batches of 64 (the property behind the collective-free multi-GPU sharding).
"""
import numpy as np
import pytest
import torch

from cmdgen_amd import hip_backend
from cmdgen_amd.synthetic import ModelConfig, make_state_dict, make_pockets
from bench import bounded_config

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda')


def eval_inputs(pb, cfg, seed=12345):
    """Phar points inside the pocket (the geometry a trained model holds: every pocket node within a few hops of a moving node)."""
    B = len(pb.size)
    rng = np.random.Generator(np.random.PCG64(seed))
    nl = int(pb.num_nodes_phar.sum())
    pm = np.repeat(np.arange(B), pb.num_nodes_phar)
    starts = np.concatenate([[0], np.cumsum(pb.size)[:-1]])
    com = np.add.reduceat(pb.x.astype(np.float64), starts, axis=0) / pb.size[:, None]
    v = rng.normal(size=(nl, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    xin = (com[pm] + v * 5.0 * np.cbrt(rng.uniform(size=(nl, 1)))).astype(np.float32)
    xh = np.concatenate([xin, rng.normal(size=(nl, cfg.phar_nf)).astype(np.float32)], 1)
    xq = np.concatenate([pb.x, pb.one_hot / cfg.norm_values[1]], 1).astype(np.float32)
    t = rng.uniform(0.05, 0.95, size=B).astype(np.float32)
    return xh, xq, t


def handle_for(cfg, sd, pb):
    h = hip_backend.Handle(cfg.as_dict(), 0)
    h.load_state_dict(sd)
    h.set_layout(pb.num_nodes_phar, pb.size)
    return h


def forward(h, xh, xq, t):
    eps, _ = h.dynamics_forward(torch.from_numpy(xh).to(DEV), torch.from_numpy(xq).to(DEV), torch.from_numpy(t).to(DEV))
    return eps.cpu().numpy()


SHAPES = [('CA', 64), ('CA', 256), ('full-atom', 256)]


@pytest.mark.parametrize('rep, B', SHAPES)
def test_reflection_and_rotation_equivariance_at_full_size(rep, B):
    cfg = ModelConfig(residue_nf=20 if rep == 'CA' else 11, timesteps=1000)
    sd = make_state_dict(cfg, seed=0)
    pb = make_pockets(B, rep)
    h = handle_for(cfg, sd, pb)
    xh, xq, t = eval_inputs(pb, cfg)
    e0 = forward(h, xh, xq, t)
    e0b = forward(h, xh, xq, t)
    c = h.counters()
    assert c['edges'] // c['evaluations'] > (400 if rep == 'CA' else 10000) * B          # the full-size graph, phar points inside the pocket
    assert np.isfinite(e0).all()
    scale = float(np.abs(e0).max())
    noise = float(np.abs(e0 - e0b).max())                                                # run to run: float atomics of tile-boundary segments
    assert noise <= 2e-6 * scale
    # ---- reflection: bit-identical graph and invariants
    xh_r, xq_r = xh.copy(), xq.copy()
    xh_r[:, 0] *= -1.0; xq_r[:, 0] *= -1.0
    e1 = forward(h, xh_r, xq_r, t)
    want_r = e0.copy(); want_r[:, 0] *= -1.0
    assert float(np.abs(e1 - want_r).max()) <= max(4.0 * noise, 2e-6 * scale)
    # ---- a general rotation and a translation
    rng = np.random.Generator(np.random.PCG64(7))
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] *= -1.0
    shift = np.array([3.0, -2.0, 1.5])
    xh_q, xq_q = xh.copy(), xq.copy()
    xh_q[:, :3] = (xh[:, :3].astype(np.float64) @ q.T + shift).astype(np.float32)
    xq_q[:, :3] = (xq[:, :3].astype(np.float64) @ q.T + shift).astype(np.float32)
    e2 = forward(h, xh_q, xq_q, t)
    want = e0.copy(); want[:, :3] = (e0[:, :3].astype(np.float64) @ q.T).astype(np.float32)
    err = np.abs(e2 - want).max(axis=1)
    # rounding of the rotated inputs (1 ulp of |x| ~ 1e-6 A) moves every distance a little; rows next to a pair that crossed the cutoff move more
    # (an equivariance bug shows as O(1) relative errors)
    assert float(np.median(err)) <= 1e-4 * scale, (float(np.median(err)), scale)
    assert float(np.mean(err <= 1e-3 * scale)) >= 0.9, (float(np.mean(err <= 1e-3 * scale)), float(err.max()), scale)
    print(f'[{rep} B={B}] edges/evaluation {c["edges"] // c["evaluations"]}, max|eps| {scale:.3g}; run to run {noise / scale:.1e}; reflection '
          f'{float(np.abs(e1 - want_r).max()) / scale:.1e}; rotation + translation: median {float(np.median(err)) / scale:.1e}, max {float(err.max()) / scale:.1e} (relative to max|eps|)')
    h.close()


@pytest.mark.parametrize('rep, B', SHAPES)
def test_a_pockets_result_does_not_depend_on_its_batch(rep, B):
    cfg = ModelConfig(residue_nf=20 if rep == 'CA' else 11, timesteps=1000)
    sd = make_state_dict(cfg, seed=0)
    pb = make_pockets(B, rep, ragged=True)                 # ragged: tile and chunk boundaries of the permuted batch fall elsewhere
    xh, xq, t = eval_inputs(pb, cfg)
    h = handle_for(cfg, sd, pb)
    e0 = forward(h, xh, xq, t)
    e0b = forward(h, xh, xq, t)
    h.close()
    scale, noise = float(np.abs(e0).max()), float(np.abs(e0 - e0b).max())
    # the same pockets in reversed order
    perm = np.arange(B)[::-1]
    ps, pe = np.concatenate([[0], np.cumsum(pb.num_nodes_phar)]), np.concatenate([[0], np.cumsum(pb.size)])
    rows_p = np.concatenate([np.arange(ps[b], ps[b + 1]) for b in perm])
    rows_q = np.concatenate([np.arange(pe[b], pe[b + 1]) for b in perm])
    h2 = hip_backend.Handle(cfg.as_dict(), 0)
    h2.load_state_dict(sd)
    h2.set_layout(pb.num_nodes_phar[perm], pb.size[perm])
    e1 = forward(h2, np.ascontiguousarray(xh[rows_p]), np.ascontiguousarray(xq[rows_q]), np.ascontiguousarray(t[perm]))
    assert float(np.abs(e1 - e0[rows_p]).max()) <= max(8.0 * noise, 4e-6 * scale)
    # the first quarter of the batch alone
    nb = B // 4
    h2.set_layout(pb.num_nodes_phar[:nb], pb.size[:nb])
    e2 = forward(h2, np.ascontiguousarray(xh[:ps[nb]]), np.ascontiguousarray(xq[:pe[nb]]), np.ascontiguousarray(t[:nb]))
    assert float(np.abs(e2 - e0[:ps[nb]]).max()) <= max(8.0 * noise, 4e-6 * scale)
    print(f'[{rep} B={B} ragged] run to run {noise / scale:.1e}; reversed batch {float(np.abs(e1 - e0[rows_p]).max()) / scale:.1e}; first quarter alone '
          f'{float(np.abs(e2 - e0[:ps[nb]]).max()) / scale:.1e} (relative to max|eps|)')
    h2.close()


def test_a_pockets_chain_does_not_depend_on_its_batch_at_256_pockets():
    """256 pockets sampled together against the same pockets in four batches of 64: K = 20 steps of the bounded-schedule model (the
    bench's headline model: phar points stay inside the pocket, every step evaluates the full graph)."""
    cfg = bounded_config(20, 1000)          # (residue_nf, timesteps)
    sd = make_state_dict(cfg, seed=0)
    K = 20

    def chain(lo, hi):
        sub = make_pockets(hi - lo, 'CA', first_index=lo)
        h = handle_for(cfg, sd, sub)
        x, _, _ = h.sample_chain(torch.from_numpy(sub.x).to(DEV), torch.from_numpy(sub.one_hot).to(DEV), K, seed=5, pocket_ids=sub.pocket_index)
        st = h.chain_status()
        h.close()
        assert st['nan_resets'] == 0
        return x.cpu().numpy()

    whole = chain(0, 256)
    parts = np.concatenate([chain(lo, lo + 64) for lo in range(0, 256, 64)])
    assert np.array_equal(whole[:, 3:], parts[:, 3:])                                   # one-hot types
    # coordinates: the same arithmetic per pocket; only the float atomics of tile-boundary segments (and which segments those are) differ
    assert float(np.abs(whole[:, :3] - parts[:, :3]).max()) <= 1e-4
    assert float(np.sqrt(np.mean((whole[:, :3] - parts[:, :3]) ** 2))) <= 1e-5
    print(f'[256 pockets at once vs 4 x 64, K = {K}] coordinates: max {float(np.abs(whole[:, :3] - parts[:, :3]).max()):.1e} A, '
          f'RMS {float(np.sqrt(np.mean((whole[:, :3] - parts[:, :3]) ** 2))):.1e} A; types identical')


@pytest.mark.parametrize('rep, B, K', [('CA', 64, 30), ('full-atom', 8, 30), ('full-atom', 64, 6)])
def test_chains_are_reproducible_bit_for_bit(rep, B, K):
    """Two fresh handles, the same Philox seed: identical bits, step by step.  Per receiver a segment sum is ordered inside a tile and has at
    most two float-atomic partials across tiles (which commute) as long as the receiver's edges are fewer than a tile's rows: C-alpha layouts
    have ~20 edges per receiver on 32-row tiles, dense (full-atom) layouts ~60 on the 128-row kernels with >= 128-row chunks (pick_tiles).
    The chain starts with every phar point at the pocket centre - the densest graph it sees."""
    cfg = bounded_config(20 if rep == 'CA' else 11, 1000)
    sd = make_state_dict(cfg, seed=0)
    pb = make_pockets(B, rep)
    outs = []
    for _ in range(2):
        h = handle_for(cfg, sd, pb)
        x, xp, zs = h.sample_chain(torch.from_numpy(pb.x).to(DEV), torch.from_numpy(pb.one_hot).to(DEV), K, seed=9, pocket_ids=pb.pocket_index, want_steps=True)
        outs.append((x.cpu().numpy(), xp.cpu().numpy(), zs.cpu().numpy()))
        h.close()
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
