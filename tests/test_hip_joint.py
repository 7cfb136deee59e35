"""GPU parity tests for the JOINT model (mode 'joint'): EGNNDynamics with update_pocket_coords=True,
EnVariationalDiffusion.sample and .inpaint (RePaint) through the C ABI, against
(a) the G9 golden vectors captured from the reference and (b) the oracle on seeded inputs.

Tolerances as in test_hip_parity.py: one evaluation 2e-5 relative; chains with injected noise
coordinate RMS <= 1e-4 * max(1, max|x|), types exact.
"""
import numpy as np
import pytest
import torch

from helpers import (load_golden, rms, JointNoiseTape, joint_cfg, joint_cases, joint_inpaint_case)
from oracle import ref_cpu
from cmdgen_amd import hip_backend
from cmdgen_amd.synthetic import ModelConfig, make_state_dict, make_pockets, min_cutoff_margin

pytestmark = pytest.mark.gpu

G9 = load_golden('g9_joint.npz')
EVAL_TOL = 2e-5
_handles = {}


def handle_for(cfg, seed, gain=1.0):
    key = (tuple(sorted((k, str(v)) for k, v in cfg.as_dict().items())), seed, gain)
    if key not in _handles:
        h = hip_backend.Handle(cfg.as_dict(), 0)
        h.load_state_dict(make_state_dict(cfg, seed=seed, coord_gain=gain))
        _handles[key] = h
    return _handles[key]


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def gpu_ok(name):
    return '_h32_' not in name          # hidden_nf 32 is below the 64-column wave tile (oracle-only fixtures)


# ------------------------------------------------------------------ one evaluation
@pytest.mark.parametrize('name', [n for n in joint_cases(G9, 'dyn') if gpu_ok(n)])
def test_joint_dynamics_matches_reference(name):
    k = f'dyn/{name}/'
    H, L, B, R, seed, first = [int(v) for v in G9[k + 'meta']]
    cfg = joint_cfg(H, L, R)
    h = handle_for(cfg, seed)
    pm, qm = G9[k + 'phar_mask'], G9[k + 'pocket_mask']
    h.set_layout(np.bincount(pm, minlength=B), np.bincount(qm, minlength=B))
    ep, eq = h.dynamics_forward(dev(G9[k + 'xh_phar']), dev(G9[k + 'xh_pocket']), dev(G9[k + 't']))
    torch.cuda.synchronize()
    for got, want in ((ep.cpu().numpy(), G9[k + 'eps_phar']), (eq.cpu().numpy(), G9[k + 'eps_pocket'])):
        assert np.abs(got - want).max() <= EVAL_TOL * max(1.0, np.abs(want).max())
    # the velocity is COM-free over all nodes of each sample (dynamics.py:133-136)
    vel = np.concatenate([ep.cpu().numpy()[:, :3], eq.cpu().numpy()[:, :3]])
    m = np.concatenate([pm, qm])
    for b in range(B):
        assert np.abs(vel[m == b].sum(0)).max() < 1e-4


def test_joint_dynamics_fuzz_vs_oracle():
    """ragged batches, several widths/depths, against the oracle"""
    for it, (H, L, B) in enumerate([(64, 1, 1), (64, 3, 5), (128, 2, 3), (256, 5, 4)]):
        cfg = joint_cfg(H, L)
        seed = 900 + it
        h = handle_for(cfg, seed)
        p = ref_cpu.to_torch_params(make_state_dict(cfg, seed=seed, coord_gain=1.0))
        first = 500000 + 1000 * it
        while True:
            pb = make_pockets(B, 'CA', ragged=True, first_index=first)
            rng = np.random.Generator(np.random.PCG64(first))
            nl = pb.num_nodes_phar
            pm = np.repeat(np.arange(B), nl)
            com = np.stack([pb.x[pb.mask == b].mean(0) for b in range(B)])
            xp = (com[pm] + rng.normal(size=(len(pm), 3)) * 3.0).astype(np.float32)
            xq = (pb.x + rng.normal(size=pb.x.shape) * 0.3).astype(np.float32)
            if min_cutoff_margin(np.concatenate([xp, xq]), np.concatenate([pm, pb.mask]), 6.0) > 2e-3:
                break
            first += 7
        xh_phar = np.concatenate([xp, rng.normal(size=(len(pm), 8)).astype(np.float32)], 1)
        xh_pocket = np.concatenate([xq, (pb.one_hot / 4 + rng.normal(size=pb.one_hot.shape) * 0.2).astype(np.float32)], 1)
        t = rng.uniform(size=(B, 1)).astype(np.float32)
        with torch.no_grad():
            wp, wq = ref_cpu.dynamics_forward(p, cfg.as_dict(), torch.from_numpy(xh_phar), torch.from_numpy(xh_pocket),
                                              torch.from_numpy(t), torch.from_numpy(pm), torch.from_numpy(pb.mask))
        h.set_layout(nl, pb.size)
        ep, eq = h.dynamics_forward(dev(xh_phar), dev(xh_pocket), dev(t))
        for got, want in ((ep.cpu().numpy(), wp.numpy()), (eq.cpu().numpy(), wq.numpy())):
            assert np.abs(got - want).max() <= EVAL_TOL * max(1.0, np.abs(want).max()), (H, L, B)


def test_joint_handle_needs_pocket_output_and_refuses_conditional_chain():
    cfg = joint_cfg(64, 1)
    h = handle_for(cfg, 7)
    h.set_layout([3], [5])
    with pytest.raises(hip_backend.CmdgenError, match='eps_pocket'):
        h.dynamics_forward(torch.zeros(3, 11).cuda(), torch.zeros(5, 23).cuda(), torch.zeros(1).cuda(), want_pocket=False)
    with pytest.raises(hip_backend.CmdgenError, match='cmdgen_joint_chain'):
        h.sample_chain(torch.zeros(5, 3).cuda(), torch.zeros(5, 20).cuda(), 5)
    n_steps, n_draws = h.joint_plan(5, 1, 1, False)
    short = torch.zeros(n_draws - 1, 3 * 11 + 5 * 23).cuda()
    xo, po = torch.zeros(3, 11).cuda(), torch.zeros(5, 23).cuda()
    rc = h.lib.cmdgen_joint_chain(h.h, None, None, None, None, None, None, 5, 1, 1, hip_backend._ptr(short), n_draws - 1,
                                  0, None, hip_backend._ptr(xo), hip_backend._ptr(po), None, 0, h._stream())
    assert rc != 0 and b'combined draws' in h.lib.cmdgen_last_error(h.h)
    # a conditional handle refuses joint chains
    hc = hip_backend.Handle(ModelConfig(hidden_nf=64, n_layers=1).as_dict(), 0)
    hc.load_state_dict(make_state_dict(ModelConfig(hidden_nf=64, n_layers=1), seed=1))
    hc.set_layout([3], [5])
    with pytest.raises(hip_backend.CmdgenError, match='update_pocket_coords'):
        hc.joint_chain(5)


def test_joint_plan_matches_repaint_schedule():
    cfg = joint_cfg(64, 1)
    h = handle_for(cfg, 7)
    for key, want in G9.items():
        if key.startswith('schedule/'):
            r, j, T = [int(s[1:]) for s in key.split('/')[1].split('_')]
            n_steps, n_draws = h.joint_plan(T, r, j, True)
            assert n_steps == int(want.sum())
            assert n_draws == 2 + 2 * n_steps + (len(want) - 1)
            n_steps2, n_draws2 = h.joint_plan(T, r, j, False)       # plain sampling ignores the schedule
            assert n_steps2 == T and n_draws2 == T + 2


# ------------------------------------------------------------------ EnVariationalDiffusion.sample
@pytest.mark.parametrize('use_graph', [False, True])
@pytest.mark.parametrize('name', [n for n in joint_cases(G9, 'sample') if gpu_ok(n)])
def test_joint_sample_matches_reference(name, use_graph):
    k = f'sample/{name}/'
    H, L, B, R, seed, K = [int(v) for v in G9[k + 'meta']]
    cfg = joint_cfg(H, L, R)
    h = handle_for(cfg, seed)
    nl, npk = G9[k + 'num_phar'], G9[k + 'num_pocket']
    h.set_layout(nl, npk)
    xh_phar, xh_pocket, z_steps = h.joint_chain(K, noise=dev(G9[k + 'noise']), want_steps=True, use_graph=use_graph)
    st = h.chain_status()
    z = z_steps.cpu().numpy()
    want = G9[k + 'z_steps']
    assert z.shape == want.shape
    for s in range(K):
        assert np.abs(z[s] - want[s]).max() < 1e-4 * max(1.0, np.abs(want[s]).max()), s
    for got, ref in ((xh_phar.cpu().numpy(), G9[k + 'xh_phar']), (xh_pocket.cpu().numpy(), G9[k + 'xh_pocket'])):
        assert np.array_equal(got[:, 3:], ref[:, 3:])
        assert rms(got[:, :3], ref[:, :3]) <= 1e-4 * max(1.0, np.abs(ref[:, :3]).max())
    assert st['max_rel_com_error'] < 1e-2 and st['nan_resets'] == 0


# ------------------------------------------------------------------ EnVariationalDiffusion.inpaint
def run_inpaint(h, phar, pocket, fp, fq, K, r, j, noise, use_graph, want_steps=False):
    h.set_layout(phar['size'], pocket['size'])
    return h.joint_chain(K, phar=(dev(phar['x']), dev(phar['one_hot'])), pocket=(dev(pocket['x']), dev(pocket['one_hot'])),
                         phar_fixed=dev(fp), pocket_fixed=dev(fq), resamplings=r, jump_length=j,
                         noise=None if noise is None else dev(noise), use_graph=use_graph, want_steps=want_steps)


@pytest.mark.parametrize('use_graph', [False, True])
@pytest.mark.parametrize('name', [n for n in joint_cases(G9, 'inpaint') if gpu_ok(n)])
def test_joint_inpaint_matches_reference(name, use_graph):
    cfg, sd, phar, pocket, K, r, j = joint_inpaint_case(G9, name)
    k = f'inpaint/{name}/'
    seed = int(G9[k + 'meta'][4])
    h = handle_for(cfg, seed)
    xh_phar, xh_pocket, _ = run_inpaint(h, phar, pocket, G9[k + 'phar_fixed'], G9[k + 'pocket_fixed'], K, r, j,
                                        G9[k + 'noise'], use_graph)
    st = h.chain_status()
    for got, ref in ((xh_phar.cpu().numpy(), G9[k + 'xh_phar']), (xh_pocket.cpu().numpy(), G9[k + 'xh_pocket'])):
        assert np.array_equal(got[:, 3:], ref[:, 3:])
        assert rms(got[:, :3], ref[:, :3]) <= 1e-4 * max(1.0, np.abs(ref[:, :3]).max())
    assert st['max_rel_com_error'] < 1e-2


@pytest.mark.parametrize('K,r,j', [(6, 2, 2), (7, 3, 1), (8, 2, 3), (5, 1, 1)])
def test_joint_inpaint_schedules_vs_oracle(K, r, j):
    """jump-back schedules (resamplings > 1, jump_length > 1) against the oracle, z after every step"""
    cfg = joint_cfg(64, 2)
    seed = 77
    h = handle_for(cfg, seed)
    p = ref_cpu.to_torch_params(make_state_dict(cfg, seed=seed, coord_gain=1.0))
    B = 3
    first = 880000 + 100 * K + 10 * r + j
    for attempt in range(200):       # small pockets: the search for a seed whose every evaluation keeps clear of the cutoff is cheap
        pb = make_pockets(B, 'CA', n_pocket_nodes=14, n_phar=6, first_index=first)
        nl = pb.num_nodes_phar
        pm = np.repeat(np.arange(B, dtype=np.int64), nl)
        rng = np.random.Generator(np.random.PCG64(first))
        com = np.stack([pb.x[pb.mask == b].mean(0) for b in range(B)])
        phar = {'x': (com[pm] + rng.normal(size=(len(pm), 3)) * 2.0).astype(np.float32),
                'one_hot': np.eye(8, dtype=np.float32)[rng.integers(0, 8, size=len(pm))], 'size': nl, 'mask': pm}
        pocket = {'x': pb.x, 'one_hot': pb.one_hot, 'size': pb.size, 'mask': pb.mask}
        fp = (rng.uniform(size=len(pm)) < 0.3).astype(np.float32)
        fq = np.ones(len(pb.mask), dtype=np.float32)
        n_steps, n_draws = h.joint_plan(K, r, j, True)
        Nl, Np = len(pm), len(pb.mask)
        noise = rng.normal(size=(n_draws, Nl * 11 + Np * 23)).astype(np.float32)
        tape = JointNoiseTape(noise, Nl, Np)
        margins = []
        orig = ref_cpu.get_edges

        def rec(mask, x, cutoff):
            margins.append(min_cutoff_margin(x.numpy(), mask.numpy(), cutoff))
            return orig(mask, x, cutoff)
        ref_cpu.get_edges = rec
        try:
            with torch.no_grad():
                t = lambda d: {kk: torch.from_numpy(np.asarray(v).copy()) for kk, v in d.items()}
                wp, wq, _, _, chain = ref_cpu.joint_inpaint(p, cfg.as_dict(), t(phar), t(pocket), torch.from_numpy(fp),
                                                            torch.from_numpy(fq), resamplings=r, jump_length=j,
                                                            timesteps=K, noise=tape, return_chain=True)
        finally:
            ref_cpu.get_edges = orig
        assert tape.i == n_draws
        if min(margins) > 5e-4:
            break
        first += 1
    else:
        pytest.fail('no seed with a safe cutoff margin found')
    xh_phar, xh_pocket, z_steps = run_inpaint(h, phar, pocket, fp, fq, K, r, j, noise, use_graph=True, want_steps=True)
    assert z_steps.shape[0] == n_steps
    # the oracle's chain lists the state after every merge and after every jump back; the device records merges
    sched = ref_cpu.get_repaint_schedule(r, j, K)
    merges, pos = [], 0
    for i, n in enumerate(sched):
        for jj in range(n):
            merges.append(pos); pos += 1
            if jj == n - 1 and i < len(sched) - 1:
                pos += 1
    z = z_steps.cpu().numpy()
    for step, ci in enumerate(merges):
        want = np.concatenate([chain[ci][0].numpy().ravel(), chain[ci][1].numpy().ravel()])
        assert np.abs(z[step] - want).max() < 2e-4 * max(1.0, np.abs(want).max()), (step, ci)
    for got, ref in ((xh_phar.cpu().numpy(), wp.numpy()), (xh_pocket.cpu().numpy(), wq.numpy())):
        assert np.array_equal(got[:, 3:], ref[:, 3:])
        assert rms(got[:, :3], ref[:, :3]) <= 1e-4 * max(1.0, np.abs(ref[:, :3]).max())


# ------------------------------------------------------------------ full size, device noise: properties
def test_joint_inpaint_full_size_properties_and_sharding():
    """B=64 C-alpha pockets, shipped widths, Philox noise: the known pocket comes back (types exactly,
    coordinates up to the sigma_0-scale decode noise and one rigid shift), outputs are COM-free, results are
    deterministic in the seed and independent of how pockets are sharded (keyed by global pocket id)."""
    cfg = joint_cfg(256, 5)
    h = handle_for(cfg, 5, gain=1e-3)
    B, K = 64, 10
    pb = make_pockets(B, 'CA', ragged=True, first_index=40000)
    nl = pb.num_nodes_phar
    pm = np.repeat(np.arange(B, dtype=np.int64), nl)

    def run(sl, ids):
        sel_q = np.isin(pb.mask, sl)
        sel_p = np.isin(pm, sl)
        phar = {'x': np.zeros((int(sel_p.sum()), 3), np.float32), 'one_hot': np.zeros((int(sel_p.sum()), 8), np.float32),
                'size': nl[sl]}
        pocket = {'x': pb.x[sel_q], 'one_hot': pb.one_hot[sel_q], 'size': pb.size[sl]}
        h.set_layout(phar['size'], pocket['size'])
        out = h.joint_chain(K, phar=(dev(phar['x']), dev(phar['one_hot'])), pocket=(dev(pocket['x']), dev(pocket['one_hot'])),
                            phar_fixed=dev(np.zeros(int(sel_p.sum()), np.float32)),
                            pocket_fixed=dev(np.ones(int(sel_q.sum()), np.float32)), seed=1234, pocket_ids=ids)
        st = h.chain_status()
        assert st['max_rel_com_error'] < 1e-2
        return out[0].cpu().numpy(), out[1].cpu().numpy()

    full_p, full_q = run(np.arange(B), np.arange(B))
    again_p, again_q = run(np.arange(B), np.arange(B))
    # same seed -> same chain, up to the order of the float atomics at tile boundaries (amplified by 1/alpha_T)
    scale = max(1.0, np.abs(full_p[:, :3]).max())
    assert np.array_equal(full_p[:, 3:], again_p[:, 3:]) and np.array_equal(full_q[:, 3:], again_q[:, 3:])
    assert np.abs(full_p[:, :3] - again_p[:, :3]).max() < 1e-3 * scale and np.abs(full_q[:, :3] - again_q[:, :3]).max() < 1e-3 * scale
    assert np.array_equal(full_q[:, 3:], pb.one_hot)                                    # known types survive
    assert np.all(full_p[:, 3:].sum(1) == 1)
    allx = np.concatenate([full_p[:, :3], full_q[:, :3]]); allm = np.concatenate([pm, pb.mask])
    for b in range(B):
        assert np.abs(allx[allm == b].sum(0)).max() < 5e-2 + 1e-3                       # COM-free (or drift-fixed)
        q = full_q[pb.mask == b, :3]; x0 = pb.x[pb.mask == b]
        shift = (q - x0).mean(0)
        assert np.abs(q - x0 - shift).max() < 0.05                                      # rigid copy of the known pocket
    half_p, half_q = run(np.arange(B // 2, B), np.arange(B // 2, B))
    sel = pm >= B // 2
    assert np.array_equal(half_p[:, 3:], full_p[sel][:, 3:])
    assert np.abs(half_p[:, :3] - full_p[sel][:, :3]).max() < 1e-3 * scale              # same draws whatever the sharding


# ------------------------------------------------------------------ Python API (the reference's classes)
def small_joint_module(H=64, L=2, hist=None):
    from cmdgen_amd.equivariant_diffusion.dynamics import EGNNDynamics
    from cmdgen_amd.equivariant_diffusion.en_diffusion import EnVariationalDiffusion
    dyn = EGNNDynamics(phar_nf=8, residue_nf=20, n_dims=3, joint_nf=32, hidden_nf=H, n_layers=L, attention=True,
                       tanh=True, norm_constant=1, inv_sublayers=1, sin_embedding=False, normalization_factor=100,
                       aggregation_method='sum', edge_cutoff=6.0, update_pocket_coords=True)
    ddpm = EnVariationalDiffusion(dynamics=dyn, phar_nf=8, residue_nf=20, n_dims=3, timesteps=500,
                                  noise_schedule='polynomial_2', noise_precision=1e-5, loss_type='l2',
                                  norm_values=[1, 4], size_histogram=np.ones((30, 70)) if hist is None else hist)
    return ddpm


def test_python_api_joint_sample_and_inpaint_match_oracle():
    name = 'ji_h64_K5_r2j1_partial'
    cfg, sd, phar, pocket, K, r, j = joint_inpaint_case(G9, name)
    k = f'inpaint/{name}/'
    ddpm = small_joint_module(64, 2)
    ddpm.load_state_dict({kk[len('ddpm.'):]: torch.from_numpy(v) for kk, v in sd.items()}, strict=True)
    ddpm = ddpm.cuda()
    t = lambda d: {kk: torch.from_numpy(np.asarray(v).copy()).cuda() for kk, v in d.items()}
    xh_phar, xh_pocket, pm, qm = ddpm.inpaint(t(phar), t(pocket), dev(G9[k + 'phar_fixed']), dev(G9[k + 'pocket_fixed']),
                                              resamplings=r, jump_length=j, timesteps=K, noise=dev(G9[k + 'noise']))
    assert np.array_equal(xh_phar.cpu().numpy()[:, 3:], G9[k + 'xh_phar'][:, 3:])
    assert rms(xh_phar.cpu().numpy()[:, :3], G9[k + 'xh_phar'][:, :3]) < 1e-4 * max(1.0, np.abs(G9[k + 'xh_phar'][:, :3]).max())
    assert rms(xh_pocket.cpu().numpy()[:, :3], G9[k + 'xh_pocket'][:, :3]) < 1e-4 * max(1.0, np.abs(G9[k + 'xh_pocket'][:, :3]).max())
    assert torch.equal(pm.cpu(), torch.from_numpy(phar['mask'])) and torch.equal(qm.cpu(), torch.from_numpy(pocket['mask']))
    assert ddpm.get_repaint_schedule(2, 2, 6) == G9['schedule/r2_j2_T6'].tolist()
    # frames: jump_length == 1 allows return_frames > 1; frame 0 is the final sample
    fr_phar, fr_pocket, _, _ = ddpm.inpaint(t(phar), t(pocket), dev(G9[k + 'phar_fixed']), dev(G9[k + 'pocket_fixed']),
                                            resamplings=1, jump_length=1, timesteps=6, return_frames=3, seed=3)
    assert fr_phar.shape == (3,) + tuple(xh_phar.shape) and fr_pocket.shape == (3,) + tuple(xh_pocket.shape)
    assert torch.all(fr_phar[0][:, 3:].sum(1) == 1) and fr_phar[1:].abs().sum() > 0
    # unconditional joint sampling with device noise: shapes, one-hot types, COM-free
    xs, qs, m1, m2 = ddpm.sample(3, torch.tensor([5, 7, 4]), torch.tensor([12, 9, 15]), timesteps=8, seed=11)
    assert xs.shape == (16, 11) and qs.shape == (36, 23)
    assert torch.all(xs[:, 3:].sum(1) == 1) and torch.all(qs[:, 3:].sum(1) == 1)
    allx = torch.cat([xs[:, :3], qs[:, :3]]).cpu().numpy(); allm = torch.cat([m1, m2]).cpu().numpy()
    for b in range(3):
        assert np.abs(allx[allm == b].sum(0)).max() < 5e-2 + 1e-3
    xs2, qs2, _, _ = ddpm.sample(3, torch.tensor([5, 7, 4]), torch.tensor([12, 9, 15]), timesteps=8, seed=11)
    sc = float(max(1.0, xs[:, :3].abs().max(), qs[:, :3].abs().max()))
    assert torch.equal(xs[:, 3:], xs2[:, 3:]) and torch.equal(qs[:, 3:], qs2[:, 3:])
    assert float((xs - xs2).abs().max()) < 1e-3 * sc and float((qs - qs2).abs().max()) < 1e-3 * sc   # up to float-atomic order


def test_generate_phars_joint_mode_end_to_end(tmp_path):
    """mode 'joint': PDB -> pocket -> RePaint inpainting with every pocket node fixed (lightning_modules.py:466-486)
    -> 'Molecule_k' dict, through PharPocketDDPM.generate_phars and the CLI (--resamplings / --jump_length)."""
    import json
    from argparse import Namespace
    from test_hip_parity import PDB_TEXT
    from cmdgen_amd.lightning_modules import PharPocketDDPM
    from cmdgen_amd import generate_phars as cli
    hp = dict(outdir='out', dataset='crossdock', datadir='data', batch_size=4, lr=1e-4,
              egnn_params=Namespace(device='cuda', edge_cutoff=6.0, joint_nf=32, hidden_nf=128, n_layers=3,
                                    attention=True, tanh=True, norm_constant=1, inv_sublayers=1, sin_embedding=False,
                                    aggregation_method='sum', normalization_factor=100),
              diffusion_params=Namespace(diffusion_steps=500, diffusion_noise_schedule='polynomial_2',
                                         diffusion_noise_precision=1e-5, diffusion_loss_type='l2',
                                         normalize_factors=[1, 4]),
              num_workers=0, augment_noise=0, augment_rotation=False, clip_grad=True, eval_epochs=50,
              eval_params=Namespace(n_eval_samples=100, eval_batch_size=100), mode='joint',
              node_histogram=np.ones((30, 70)), pocket_representation='CA')
    model = PharPocketDDPM(**hp)
    sd = make_state_dict(ModelConfig(hidden_nf=128, n_layers=3), seed=0)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    ck = tmp_path / 'j.ckpt'
    model.save_checkpoint(str(ck))
    pdb = tmp_path / 'p.pdb'
    pdb.write_text(PDB_TEXT)
    model = PharPocketDDPM.load_from_checkpoint(str(ck), map_location='cuda').cuda()
    assert model.mode == 'joint' and type(model.ddpm).__name__ == 'EnVariationalDiffusion'
    ids = [f'A:{i}' for i in range(1, 7)]
    out = model.generate_phars(str(pdb), 2, pocket_ids=ids, num_nodes_phar=torch.tensor([5, 5]), timesteps=20,
                               resamplings=2, jump_length=2, seed=3)
    assert sorted(out) == [f'Molecule_{k}' for k in range(1, 6)]
    pts = [c for feats in out.values() for cs in feats.values() for c in cs]
    assert len(pts) == 10 and all(torch.isfinite(c).all() for c in pts)
    st = model.ddpm.last_chain_status
    assert st['max_rel_com_error'] < 1e-2
    plain = cli.main([str(ck), '--pdbfile', str(pdb), '--resi_list'] + ids +
                     ['--n_samples', '3', '--num_nodes_phar', '4', '--timesteps', '10', '--resamplings', '2',
                      '--jump_length', '1', '--outdir', str(tmp_path)])
    written = json.load(open(tmp_path / cli.DEFAULT_JSON))
    assert written == plain and sorted(written) == ['Molecule_1', 'Molecule_2', 'Molecule_3', 'Molecule_4']


@pytest.mark.parametrize('mode', ['train', 'eval'])
def test_joint_loss_terms_match_reference(mode):
    """EnVariationalDiffusion.forward (en_diffusion.py:332-465): the 12 loss terms + info with t_int and every
    Gaussian draw pinned (G9 loss case, t = 0 and t = T included), evaluation on the GPU; <= 2e-5 relative."""
    from helpers import joint_loss_case, LOSS_NAMES
    cfg, sd, phar, pocket, hist = joint_loss_case(G9)
    ddpm = small_joint_module(64, 2, hist)
    ddpm.load_state_dict({kk[len('ddpm.'):]: torch.from_numpy(v) for kk, v in sd.items()}, strict=True)
    ddpm = ddpm.cuda()
    ddpm.train() if mode == 'train' else ddpm.eval()
    Nl, Np = len(phar['mask']), len(pocket['mask'])
    eps = [(dev(row[:Nl * 11].reshape(Nl, 11)), dev(row[Nl * 11:].reshape(Np, 23))) for row in G9[f'loss/{mode}/noise']]
    cu = lambda d: {kk: v.cuda() for kk, v in d.items()}
    terms = ddpm(cu(phar), cu(pocket), return_info=True, t_int=dev(G9['loss/t_int']), eps=eps)
    for n, v in zip(LOSS_NAMES, terms[:-1]):
        want = G9[f'loss/{mode}/{n}']
        got = np.asarray(v.detach().cpu().numpy() if torch.is_tensor(v) else v, dtype=np.float32)
        assert got.shape == want.shape, n
        assert np.abs(got - want).max() <= 2e-5 * max(1.0, np.abs(want).max()), (n, got, want)
    for kk, v in terms[-1].items():
        assert abs(float(v) - float(G9[f'loss/{mode}/info_{kk}'])) < 2e-5, kk
