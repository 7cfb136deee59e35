"""GPU parity tests: the HIP path (through the C ABI, via cmdgen_amd.hip_backend) against
(a) the golden vectors captured from the reference and (b) the oracle on seeded inputs.

Tolerances (fp32, stated per test):
  * one network evaluation: max |eps - eps_ref| <= 2e-5 * max(1, max|eps_ref|)
    (fp32 re-association: first-layer factorisation P_i + Q_j, MFMA k-order, v_exp/v_rcp SiLU)
  * chains with injected noise: coordinate RMS <= 1e-4 * max(1, max|x|) (north-star bound),
    one-hot types exact; fixtures keep every pair >= 2e-3 A away from the 6 A cutoff because
    the radius graph is a hard threshold (hard part 2).
"""
import numpy as np
import pytest
import torch

from helpers import (load_golden, cases_of, dynamics_case, chain_case, rms, NoiseTape)
from cmdgen_amd import hip_backend
from cmdgen_amd.synthetic import ModelConfig, make_state_dict, make_pockets, min_cutoff_margin

pytestmark = pytest.mark.gpu

EVAL_TOL = 2e-5
_handles = {}


def handle_for(cfg, sd_key, sd):
    """One handle per (config, weight set); reused across tests."""
    key = (tuple(sorted((k, str(v)) for k, v in cfg.as_dict().items())), sd_key)
    if key not in _handles:
        h = hip_backend.Handle(cfg.as_dict(), 0)
        h.load_state_dict(sd)
        _handles[key] = h
    return _handles[key]


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def supported(name):
    return '_h32_' not in name          # hidden_nf 32 is below the 64-column wave tile


G2 = load_golden('g2_dynamics.npz')
G4 = load_golden('g4_chains.npz')


def run_eval(name):
    cfg, sd, inp = dynamics_case(G2, name)
    h = handle_for(cfg, name, sd)
    h.set_layout(G2[name + '/num_nodes_phar'], G2[name + '/pocket_size'])
    eps_phar, eps_pocket = h.dynamics_forward(dev(inp['xh_phar']), dev(inp['xh_pocket']), dev(inp['t']))
    torch.cuda.synchronize()
    return cfg, h, inp, eps_phar.cpu().numpy(), eps_pocket.cpu().numpy()


@pytest.mark.parametrize('name', [n for n in cases_of(G2) if supported(n)])
def test_radius_graph_matches_reference(name):
    cfg, h, inp, _, _ = run_eval(name)
    edges = h.get_edges()
    want = G2[name + '/edges']
    assert edges.shape == want.shape, (edges.shape, want.shape)
    assert np.array_equal(edges, want)          # same set, same (row, col) order, self loops kept


def test_radius_graph_boundary_cases():
    """G3: a pair at exactly 6.0 A is kept (<=), 6.5 is not, self loops, no cross-sample edges."""
    g = load_golden('g3_edges.npz')
    cfg = ModelConfig(hidden_nf=64, n_layers=1)
    sd = make_state_dict(cfg, seed=3)
    h = handle_for(cfg, 'g3', sd)
    # sample 0 = 30 phar nodes and no pocket nodes, sample 1 = 10 pocket nodes and no phar nodes:
    # the flat order [phar..., pocket...] is then exactly the fixture's node order.
    h.set_layout([30, 0], [0, 10])
    xh_phar = np.zeros((30, 3 + cfg.phar_nf), np.float32); xh_phar[:, :3] = g['x'][:30]
    xh_pocket = np.zeros((10, 3 + cfg.residue_nf), np.float32); xh_pocket[:, :3] = g['x'][30:]
    h.dynamics_forward(dev(xh_phar), dev(xh_pocket), dev(np.array([0.5, 0.5], np.float32)))
    assert np.array_equal(h.get_edges(), g['edges'])


@pytest.mark.parametrize('name', [n for n in cases_of(G2) if supported(n)])
def test_dynamics_forward_matches_reference(name):
    cfg, h, inp, eps_phar, eps_pocket = run_eval(name)
    want = G2[name + '/eps_phar']
    tol = EVAL_TOL * max(1.0, float(np.abs(want).max()))
    err = float(np.abs(eps_phar - want).max())
    assert err <= tol, f'{name}: max abs err {err:.3e} > {tol:.3e}'
    # conditional mode: the pocket never moves
    assert np.all(eps_pocket[:, :3] == 0)
    if name + '/eps_pocket' in G2:
        wp = G2[name + '/eps_pocket']
        assert float(np.abs(eps_pocket - wp).max()) <= EVAL_TOL * max(1.0, float(np.abs(wp).max()))
    if name + '/block0_h_phar' in G2:          # last block's h of the phar rows and final x
        nl = len(inp['mask_phar'])
        L = cfg.n_layers
        hfin = h.debug_read('h', (nl + 16) * cfg.hidden_nf).reshape(-1, cfg.hidden_nf)
        wh = G2[name + f'/block{L - 1}_h_phar']
        assert float(np.abs(hfin[:nl] - wh).max()) <= 5e-5 * max(1.0, float(np.abs(wh).max()))
        assert float(np.abs(hfin[nl:nl + 16] - G2[name + f'/block{L - 1}_h_pocket_head']).max()) <= 5e-5 * max(1.0, float(np.abs(wh).max()))


@pytest.mark.parametrize('H,L,rep', [(64, 2, 'CA'), (128, 3, 'CA'), (128, 2, 'full-atom'), (512, 2, 'CA'), (512, 2, 'full-atom')])
def test_dynamics_forward_matches_oracle_other_widths(H, L, rep):
    """Oracle-checked cases for the other supported hidden sizes (seeded inputs, ragged sizes)."""
    from oracle import ref_cpu
    cfg = ModelConfig(hidden_nf=H, n_layers=L, residue_nf=20 if rep == 'CA' else 11)
    sd = make_state_dict(cfg, seed=100 + H + L, coord_gain=1.0)
    first = 5000 + H
    while True:
        pb = make_pockets(5, rep, ragged=(rep == 'CA'), n_pocket_nodes=70 if rep != 'CA' else None,
                          n_phar=7, first_index=first)
        rng = np.random.Generator(np.random.PCG64(first))
        B = len(pb.size)
        phar_mask = np.repeat(np.arange(B), pb.num_nodes_phar)
        com = np.stack([pb.x[pb.mask == b].mean(0) for b in range(B)])
        xp = (com[phar_mask] + rng.normal(size=(len(phar_mask), 3)) * 2.5).astype(np.float32)
        allx = np.concatenate([xp, pb.x]); allm = np.concatenate([phar_mask, pb.mask])
        if min_cutoff_margin(allx, allm, 6.0) > 2e-3:
            break
        first += 1000
    xh_phar = np.concatenate([xp, rng.normal(size=(len(phar_mask), cfg.phar_nf)).astype(np.float32)], 1)
    xh_pocket = np.concatenate([pb.x, pb.one_hot / 4.0], 1).astype(np.float32)
    t = rng.uniform(0.1, 0.9, size=(B, 1)).astype(np.float32)
    p = ref_cpu.to_torch_params(sd)
    with torch.no_grad():
        want, _ = ref_cpu.dynamics_forward(p, cfg.as_dict(), torch.from_numpy(xh_phar), torch.from_numpy(xh_pocket),
                                           torch.from_numpy(t), torch.from_numpy(phar_mask), torch.from_numpy(pb.mask))
    want = want.numpy()
    h = handle_for(cfg, f'oracle{H}{L}{rep}', sd)
    h.set_layout(pb.num_nodes_phar, pb.size)
    got, _ = h.dynamics_forward(dev(xh_phar), dev(xh_pocket), dev(t))
    err = float(np.abs(got.cpu().numpy() - want).max())
    assert err <= EVAL_TOL * max(1.0, float(np.abs(want).max())), err
    if H == 512:                      # hidden_nf 512 (sampling only): every tile size of the three MFMA kernels, both engines, and a short chain
        for mt in (16, 32, 64):
            for split in (True, False):
                h2 = hip_backend.Handle(cfg.as_dict(), 0)
                h2.load_state_dict(sd); h2.set_gemm_mode(split)
                for k in ('node_mt', 'edge_mt', 'coord_mt', 'embed_mt'):
                    h2.set_option(k, mt)
                h2.set_layout(pb.num_nodes_phar, pb.size)
                g2, _ = h2.dynamics_forward(dev(xh_phar), dev(xh_pocket), dev(t))
                e2 = float(np.abs(g2.cpu().numpy() - want).max())
                assert e2 <= EVAL_TOL * max(1.0, float(np.abs(want).max())), (mt, split, e2)
                h2.close()
        K = 6
        Nl = int(pb.num_nodes_phar.sum())
        noise = np.random.Generator(np.random.PCG64(first)).normal(size=(K + 2, Nl, 3 + cfg.phar_nf)).astype(np.float32)
        pocket = {'x': torch.from_numpy(pb.x), 'one_hot': torch.from_numpy(pb.one_hot), 'size': torch.from_numpy(pb.size), 'mask': torch.from_numpy(pb.mask)}
        with torch.no_grad():
            ref = ref_cpu.sample_given_pocket(p, cfg.as_dict(), pocket, pb.num_nodes_phar, timesteps=K, noise=NoiseTape(noise))
        from test_hip_parity_r2 import host_step_table
        for use_graph in (False, True):
            h.set_step_table(K, host_step_table(cfg, K))
            x, _, _ = h.sample_chain(dev(pb.x), dev(pb.one_hot), K, noise=dev(noise), use_graph=use_graph)
            wx = ref[0].numpy()
            assert rms(x[:, :3].cpu().numpy(), wx[:, :3]) <= 1e-4 * max(1.0, float(np.abs(wx[:, :3]).max()))
            assert np.array_equal(x[:, 3:].cpu().numpy(), wx[:, 3:])
        with pytest.raises(hip_backend.CmdgenError, match='hidden_nf <= 256'):
            z = torch.zeros(8, device='cuda')
            h._check(h.lib.cmdgen_train_forward(h.h, z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), None, None), 'cmdgen_train_forward')


def run_chain(name, use_graph):
    cfg, sd, pb, K = chain_case(G4, name)
    h = handle_for(cfg, name, sd)
    h.set_layout(pb.num_nodes_phar, pb.size)
    # per-step scalars evaluated by the host exactly as the reference does (bit-identical table)
    from cmdgen_amd.equivariant_diffusion.en_diffusion import PredefinedNoiseSchedule  # noqa: F401
    from oracle import ref_cpu
    table = ref_cpu.gamma_table(cfg.noise_schedule, cfg.timesteps, cfg.noise_precision)
    coef = ref_cpu.step_coefficients(table, cfg.timesteps, K).numpy()
    g0 = table[0]
    final = np.array([[float(torch.sqrt(torch.sigmoid(g0))), float(torch.sqrt(torch.sigmoid(-g0))),
                       float(torch.exp(0.5 * g0)), 0.0]], np.float32)
    h.set_step_table(K, np.concatenate([coef, final]))
    xh_phar, xh_pocket, z_steps = h.sample_chain(dev(pb.x), dev(pb.one_hot), K, noise=dev(G4[name + '/noise']),
                                                 want_steps=True, use_graph=use_graph)
    st = h.chain_status()
    return cfg, K, xh_phar.cpu().numpy(), xh_pocket.cpu().numpy(), z_steps.cpu().numpy(), st


@pytest.mark.parametrize('use_graph', [False, True])
@pytest.mark.parametrize('name', [n for n in cases_of(G4) if supported(n)])
def test_chain_with_injected_noise_matches_reference(name, use_graph):
    cfg, K, xh_phar, xh_pocket, z_steps, st = run_chain(name, use_graph)
    want = G4[name + '/xh_phar']
    scale = max(1.0, float(np.abs(want[:, :3]).max()))
    if name + '/z_steps' in G4:
        zs = G4[name + '/z_steps']
        for k in range(K):
            e = float(np.abs(z_steps[k] - zs[k]).max())
            assert e <= 1e-4 * max(1.0, float(np.abs(zs[k]).max())), f'step {k}: {e:.3e}'
    assert rms(xh_phar[:, :3], want[:, :3]) <= 1e-4 * scale
    assert np.array_equal(xh_phar[:, 3:], want[:, 3:])
    wp = G4[name + '/xh_pocket']
    assert rms(xh_pocket, wp) <= 1e-4 * max(1.0, float(np.abs(wp).max()))
    assert st['max_rel_com_error'] < 1e-2 and st['nan_resets'] == 0


def _philox_chain(h, pb, K, ids=None, seed=1234, use_graph=True):
    h.set_layout(pb.num_nodes_phar, pb.size)
    out = h.sample_chain(dev(pb.x), dev(pb.one_hot), K, noise=None, seed=seed, pocket_ids=ids, use_graph=use_graph)
    st = h.chain_status()
    return out[0].cpu().numpy(), out[1].cpu().numpy(), st


def test_full_size_batch_properties():
    """BASELINE config shape (64 C-alpha pockets, H=256, L=5), on-device Philox noise, 25 steps:
    size-independent properties - zero phar COM per sample, valid one-hot rows, pocket rigidly
    translated, counters consistent, graph replay == eager launches."""
    cfg = ModelConfig(timesteps=500)
    sd = make_state_dict(cfg, seed=0)
    h = handle_for(cfg, 'seed0', sd)
    pb = make_pockets(64, 'CA', n_phar=15)
    K = 25
    h.set_layout(pb.num_nodes_phar, pb.size)
    h.reset_counters()
    xg, pg, st = _philox_chain(h, pb, K, use_graph=True)
    c = h.counters()
    assert c['evaluations'] == K + 1 and c['nodes'] == (K + 1) * (64 * 59)
    assert c['edges'] >= (K + 1) * 64 * 59 and 0 <= c['edges_phar'] <= c['edges']   # every node keeps its self loop; edges_phar = phar receivers minus self loops
    assert st['max_rel_com_error'] < 1e-2 and st['nan_resets'] == 0
    assert np.isfinite(xg).all()
    oh = xg[:, 3:]
    assert np.all((oh == 0) | (oh == 1)) and np.all(oh.sum(1) == 1)
    pm = np.repeat(np.arange(64), 15)
    for b in range(64):
        assert np.abs(xg[pm == b, :3].sum(0)).max() < 5e-2           # CoG drift bound the reference enforces
        shift = pg[pb.mask == b, :3] - pb.x[pb.mask == b]
        assert np.abs(shift - shift[0]).max() < 1e-3 * max(1.0, np.abs(shift).max())   # rigid translation
        assert np.array_equal(pg[pb.mask == b, 3:], pb.one_hot[pb.mask == b])
    xe, pe, _ = _philox_chain(h, pb, K, use_graph=False)
    assert np.abs(xe[:, :3] - xg[:, :3]).max() <= 1e-4 * max(1.0, np.abs(xg[:, :3]).max())
    assert np.array_equal(xe[:, 3:], xg[:, 3:])


def test_sharding_independence():
    """Pockets shard embarrassingly: two shards keyed by global pocket ids reproduce the full batch."""
    cfg = ModelConfig(timesteps=500)
    sd = make_state_dict(cfg, seed=0)
    h = handle_for(cfg, 'seed0', sd)
    K = 6
    full = make_pockets(8, 'CA', ragged=True)
    xf, pf, _ = _philox_chain(h, full, K, ids=full.pocket_index)
    a = make_pockets(4, 'CA', ragged=True, first_index=0)
    b = make_pockets(4, 'CA', ragged=True, first_index=4)
    xa, _, _ = _philox_chain(h, a, K, ids=a.pocket_index)
    xb, _, _ = _philox_chain(h, b, K, ids=b.pocket_index)
    xs = np.concatenate([xa, xb])
    assert xs.shape == xf.shape
    assert np.abs(xs[:, :3] - xf[:, :3]).max() <= 1e-4 * max(1.0, np.abs(xf[:, :3]).max())
    assert np.array_equal(xs[:, 3:], xf[:, 3:])


def test_python_interface_matches_reference_chain():
    """The reference-shaped Python API (ConditionalDDPM.sample_given_pocket) on a golden chain."""
    from cmdgen_amd.equivariant_diffusion.dynamics import EGNNDynamics
    from cmdgen_amd.equivariant_diffusion.conditional_model import ConditionalDDPM
    name = 'ca_h256_K5'
    cfg, sd, pb, K = chain_case(G4, name)
    hist = np.ones((30, 70))
    dyn = EGNNDynamics(phar_nf=cfg.phar_nf, residue_nf=cfg.residue_nf, n_dims=3, joint_nf=cfg.joint_nf,
                       hidden_nf=cfg.hidden_nf, n_layers=cfg.n_layers, attention=True, tanh=True,
                       norm_constant=1, inv_sublayers=1, sin_embedding=False, normalization_factor=100,
                       aggregation_method='sum', edge_cutoff=6.0, update_pocket_coords=False)
    ddpm = ConditionalDDPM(dynamics=dyn, phar_nf=cfg.phar_nf, residue_nf=cfg.residue_nf, n_dims=3,
                           timesteps=cfg.timesteps, noise_schedule='polynomial_2', noise_precision=1e-5,
                           loss_type='l2', norm_values=[1, 4], size_histogram=hist)
    ddpm.load_state_dict({k[len('ddpm.'):]: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    ddpm = ddpm.cuda()
    pocket = {'x': dev(pb.x), 'one_hot': dev(pb.one_hot), 'size': dev(pb.size), 'mask': dev(pb.mask)}
    xh_phar, xh_pocket, phar_mask, pocket_mask = ddpm.sample_given_pocket(
        pocket, torch.from_numpy(pb.num_nodes_phar), timesteps=K, noise=dev(G4[name + '/noise']))
    want = G4[name + '/xh_phar']
    assert np.array_equal(phar_mask.cpu().numpy(), G4[name + '/phar_mask'])
    assert rms(xh_phar[:, :3].cpu().numpy(), want[:, :3]) <= 1e-4 * max(1.0, float(np.abs(want[:, :3]).max()))
    assert np.array_equal(xh_phar[:, 3:].cpu().numpy(), want[:, 3:])
    # EGNNDynamics.forward through the module interface on the chain's last state is covered by
    # test_dynamics_forward_*; here check the module-level forward agrees with the handle-level one.
    g = G2
    c2, sd2, inp = dynamics_case(g, 'ca_h256_b3')
    dyn.load_state_dict({k[len('ddpm.dynamics.'):]: torch.from_numpy(v) for k, v in sd2.items()
                         if k.startswith('ddpm.dynamics.')})
    e1, e2 = dyn(dev(inp['xh_phar']), dev(inp['xh_pocket']), dev(inp['t']), dev(inp['mask_phar']), dev(inp['mask_pocket']))
    wantp = g['ca_h256_b3/eps_phar']
    assert float(np.abs(e1.cpu().numpy() - wantp).max()) <= EVAL_TOL * max(1.0, float(np.abs(wantp).max()))
    assert np.array_equal(dyn.get_edges().numpy(), g['ca_h256_b3/edges'].astype(np.int64))


def test_errors_are_loud():
    with pytest.raises(hip_backend.CmdgenError):
        hip_backend.Handle(ModelConfig(hidden_nf=32).as_dict(), 0)      # unsupported width
    h = hip_backend.Handle(ModelConfig(hidden_nf=64, n_layers=1).as_dict(), 0)
    with pytest.raises(hip_backend.CmdgenError):
        h.set_layout([3], [10])
        h.dynamics_forward(torch.zeros(3, 11).cuda(), torch.zeros(10, 23).cuda(), torch.zeros(1).cuda())  # no weights


PDB_TEXT = """\
ATOM      1  CA  ALA A   1      11.639   6.071  -5.147  1.00  0.00           C
ATOM      2  CA  GLY A   2      14.000   7.500  -3.000  1.00  0.00           C
ATOM      3  CA  TRP A   3      16.500   9.000  -1.000  1.00  0.00           C
ATOM      4  CA  LEU A   4      12.500   9.500  -2.000  1.00  0.00           C
ATOM      5  CA  SER A   5      10.000   8.000   0.500  1.00  0.00           C
ATOM      6  CA  ASP A   6      13.500   4.000  -1.500  1.00  0.00           C
END
"""


def test_generate_phars_end_to_end(tmp_path):
    """BASELINE config[0] plumbing on the GPU: PDB -> pocket tensors -> 50 strided steps of a T=500
    model -> per-'Molecule_k' dict, through PharPocketDDPM.generate_phars and the CLI."""
    import json
    from argparse import Namespace
    from cmdgen_amd.lightning_modules import PharPocketDDPM
    from cmdgen_amd import generate_phars as cli
    hp = dict(outdir='out', dataset='crossdock', datadir='data', batch_size=4, lr=1e-4,
              egnn_params=Namespace(device='cuda', edge_cutoff=6.0, joint_nf=32, hidden_nf=256, n_layers=5,
                                    attention=True, tanh=True, norm_constant=1, inv_sublayers=1, sin_embedding=False,
                                    aggregation_method='sum', normalization_factor=100),
              diffusion_params=Namespace(diffusion_steps=500, diffusion_noise_schedule='polynomial_2',
                                         diffusion_noise_precision=1e-5, diffusion_loss_type='l2',
                                         normalize_factors=[1, 4]),
              num_workers=0, augment_noise=0, augment_rotation=False, clip_grad=True, eval_epochs=50,
              eval_params=Namespace(n_eval_samples=100, eval_batch_size=100), mode='pocket_conditioning',
              node_histogram=np.ones((30, 70)), pocket_representation='CA')
    model = PharPocketDDPM(**hp)
    sd = make_state_dict(ModelConfig(), seed=0)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    ck = tmp_path / 'm.ckpt'
    model.save_checkpoint(str(ck))
    pdb = tmp_path / 'p.pdb'
    pdb.write_text(PDB_TEXT)
    model = PharPocketDDPM.load_from_checkpoint(str(ck), map_location='cuda').cuda()
    out = model.generate_phars(str(pdb), 1, pocket_ids=[f'A:{i}' for i in range(1, 7)],
                               num_nodes_phar=torch.tensor([8]), timesteps=50, seed=3)
    assert sorted(out) == [f'Molecule_{k}' for k in range(1, 9)]           # 8 points of the one sample
    pts = [c for feats in out.values() for cs in feats.values() for c in cs]
    assert len(pts) == 8 and all(torch.isfinite(c).all() for c in pts)
    names = {t for feats in out.values() for t in feats}
    assert names <= set(model.dataset_info['phar_decoder'])
    # same seed -> same sample; the CLI writes the JSON the next pipeline stage reads
    out2 = model.generate_phars(str(pdb), 1, pocket_ids=[f'A:{i}' for i in range(1, 7)],
                                num_nodes_phar=torch.tensor([8]), timesteps=50, seed=3)
    a = torch.stack([c for k in sorted(out) for t in sorted(out[k]) for c in out[k][t]])
    b = torch.stack([c for k in sorted(out2) for t in sorted(out2[k]) for c in out2[k][t]])
    assert torch.allclose(a, b, atol=1e-3 * max(1.0, float(a.abs().max())))
    plain = cli.main([str(ck), '--pdbfile', str(pdb), '--resi_list'] + [f'A:{i}' for i in range(1, 7)] +
                     ['--n_samples', '3', '--num_nodes_phar', '4', '--timesteps', '20', '--outdir', str(tmp_path)])
    written = json.load(open(tmp_path / cli.DEFAULT_JSON))
    assert written == plain and sorted(written) == ['Molecule_1', 'Molecule_2', 'Molecule_3', 'Molecule_4']
    assert sum(len(cs) for cs in written['Molecule_1'].values()) == 3       # k-th point of ALL 3 samples (quirk Q9)


@pytest.mark.parametrize('nph,npk', [([1], [1]), ([1, 25, 3], [60, 30, 44]), ([5, 0, 7], [20, 33, 0])])
def test_edge_case_layouts_match_oracle(nph, npk):
    """Single-node samples, ragged extremes and samples with no phar / no pocket nodes."""
    from oracle import ref_cpu
    cfg = ModelConfig(hidden_nf=128, n_layers=2)
    sd = make_state_dict(cfg, seed=77, coord_gain=1.0)
    rng = np.random.Generator(np.random.PCG64(sum(nph) * 131 + sum(npk)))
    B = len(nph)
    pm, qm = np.repeat(np.arange(B), nph), np.repeat(np.arange(B), npk)
    while True:
        xp = rng.normal(size=(len(pm), 3)).astype(np.float32) * 3.0
        xq = rng.normal(size=(len(qm), 3)).astype(np.float32) * 5.0
        if min_cutoff_margin(np.concatenate([xp, xq]), np.concatenate([pm, qm]), 6.0) > 2e-3:
            break
    xh_phar = np.concatenate([xp, rng.normal(size=(len(pm), 8)).astype(np.float32)], 1)
    oh = np.eye(20, dtype=np.float32)[rng.integers(0, 20, size=len(qm))] / 4.0
    xh_pocket = np.concatenate([xq, oh], 1).astype(np.float32)
    t = rng.uniform(0.1, 0.9, size=(B, 1)).astype(np.float32)
    p = ref_cpu.to_torch_params(sd)
    with torch.no_grad():
        want, _ = ref_cpu.dynamics_forward(p, cfg.as_dict(), torch.from_numpy(xh_phar), torch.from_numpy(xh_pocket),
                                           torch.from_numpy(t), torch.from_numpy(pm), torch.from_numpy(qm))
    h = handle_for(cfg, 'edgecases', sd)
    h.set_layout(nph, npk)
    got, _ = h.dynamics_forward(dev(xh_phar), dev(xh_pocket), dev(t))
    want = want.numpy()
    assert float(np.abs(got.cpu().numpy() - want).max()) <= EVAL_TOL * max(1.0, float(np.abs(want).max()))


@pytest.mark.parametrize('mode', ['train', 'eval'])
def test_loss_terms_match_reference(mode):
    """ConditionalDDPM.forward / PharPocketDDPM.forward (loss VALUES) with pinned t_int and Gaussian draws (G6):
    the network evaluation runs in HIP, the scalar loss algebra on the host."""
    from argparse import Namespace
    from helpers import loss_case
    from cmdgen_amd.lightning_modules import PharPocketDDPM
    g = load_golden('g6_loss.npz')
    cfg, sd, phar, pocket, hist = loss_case(g)
    hp = dict(outdir='o', dataset='crossdock', datadir='d', batch_size=4, lr=1e-4,
              egnn_params=Namespace(device='cuda', edge_cutoff=6.0, joint_nf=32, hidden_nf=cfg.hidden_nf,
                                    n_layers=cfg.n_layers, attention=True, tanh=True, norm_constant=1, inv_sublayers=1,
                                    sin_embedding=False, aggregation_method='sum', normalization_factor=100),
              diffusion_params=Namespace(diffusion_steps=500, diffusion_noise_schedule='polynomial_2',
                                         diffusion_noise_precision=1e-5, diffusion_loss_type='l2',
                                         normalize_factors=[1, 4]),
              num_workers=0, augment_noise=0, augment_rotation=False, clip_grad=True, eval_epochs=50,
              eval_params=Namespace(n_eval_samples=10, eval_batch_size=10), mode='pocket_conditioning',
              node_histogram=hist, pocket_representation='CA')
    model = PharPocketDDPM(**hp)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model = model.cuda()
    model.train() if mode == 'train' else model.eval()
    cu = lambda d: {k: v.cuda() for k, v in d.items()}
    eps = [torch.from_numpy(g['eps0']), torch.from_numpy(g['eps1'])]
    terms = model.ddpm(cu(phar), cu(pocket), return_info=True, t_int=torch.from_numpy(g['t_int']), eps=eps)
    names = ['delta_log_px', 'error_t_phar', 'error_t_pocket', 'SNR_weight', 'loss_0_x_phar', 'loss_0_x_pocket',
             'loss_0_h', 'neg_log_constants', 'kl_prior', 'log_pN', 't_int', 'xh_phar_hat']
    for n, v in zip(names, terms[:-1]):
        want = g[f'{mode}/{n}']
        got = np.asarray(v.detach().cpu().numpy() if torch.is_tensor(v) else v, dtype=np.float32)
        assert np.allclose(got, want, rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(want).max()))), (n, got, want)
    data = {'phar_coords': phar['x'], 'phar_one_hot': phar['one_hot'], 'num_phar_atoms': phar['size'],
            'phar_mask': phar['mask'].float(), 'pocket_c_alpha': pocket['x'], 'pocket_one_hot': pocket['one_hot'],
            'num_pocket_nodes': pocket['size'], 'pocket_mask': pocket['mask'].float()}      # collate_fn masks are float (Q13)
    nll, info = model(data, t_int=torch.from_numpy(g['t_int']), eps=eps)
    from oracle import ref_cpu
    p = ref_cpu.to_torch_params(sd)
    with torch.no_grad():
        oterms = ref_cpu.ddpm_forward(p, cfg.as_dict(), phar, pocket, torch.from_numpy(g['t_int']), eps, mode == 'train', hist)
        onll = ref_cpu.nll_from_terms(oterms, cfg.as_dict(), phar['size'], pocket['size'], mode == 'train')
    assert np.allclose(nll.cpu().numpy(), onll.numpy(), rtol=1e-4, atol=1e-3)
    assert set(info) >= {'error_t_phar', 'SNR_weight', 'loss_0', 'kl_prior', 'log_pN', 'eps_hat_phar_x'}


def test_nan_reset_is_batch_global_like_reference():
    """Quirk Q6 (dynamics.py:129-131): one NaN coordinate resets the velocity of the WHOLE batch to zero while the
    decoded features are still produced; checked against the oracle."""
    from oracle import ref_cpu
    name = 'ca_h256_b3'
    cfg, sd, inp = dynamics_case(G2, name)
    h = handle_for(cfg, name, sd)
    h.set_layout(G2[name + '/num_nodes_phar'], G2[name + '/pocket_size'])
    xh_phar = inp['xh_phar'].copy()
    xh_phar[2, 1] = np.nan
    p = ref_cpu.to_torch_params(sd)
    with torch.no_grad():
        want, _ = ref_cpu.dynamics_forward(p, cfg.as_dict(), torch.from_numpy(xh_phar), torch.from_numpy(inp['xh_pocket']),
                                           torch.from_numpy(inp['t']), torch.from_numpy(inp['mask_phar']),
                                           torch.from_numpy(inp['mask_pocket']))
    want = want.numpy()
    h.reset_counters()
    got, _ = h.dynamics_forward(dev(xh_phar), dev(inp['xh_pocket']), dev(inp['t']))
    got = got.cpu().numpy()
    assert np.all(want[:, :3] == 0) and np.all(got[:, :3] == 0)             # every sample's velocity is reset
    ok = np.isfinite(want[:, 3:]).all(1)
    assert ok.sum() >= len(ok) - 1
    assert float(np.abs(got[ok, 3:] - want[ok, 3:]).max()) <= EVAL_TOL * max(1.0, float(np.abs(want[ok, 3:]).max()))
    assert h.counters()['nan_resets'] == 1


def test_return_frames_api():
    """return_frames > 1 (conditional_model.py:439-442, :460-465): frame 0 is the final sample, the others are
    un-normalised intermediate states with the rigidly translated pocket."""
    from cmdgen_amd.equivariant_diffusion.dynamics import EGNNDynamics
    from cmdgen_amd.equivariant_diffusion.conditional_model import ConditionalDDPM
    cfg = ModelConfig(hidden_nf=64, n_layers=1)
    dyn = EGNNDynamics(phar_nf=8, residue_nf=20, n_dims=3, joint_nf=32, hidden_nf=64, n_layers=1, attention=True,
                       tanh=True, norm_constant=1, inv_sublayers=1, normalization_factor=100, aggregation_method='sum',
                       edge_cutoff=6.0, update_pocket_coords=False)
    ddpm = ConditionalDDPM(dynamics=dyn, phar_nf=8, residue_nf=20, n_dims=3, timesteps=500,
                           noise_schedule='polynomial_2', noise_precision=1e-5, loss_type='l2', norm_values=[1, 4],
                           size_histogram=np.ones((30, 70)))
    sd = make_state_dict(cfg, seed=9)
    ddpm.load_state_dict({k[len('ddpm.'):]: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    ddpm = ddpm.cuda()
    pb = make_pockets(3, 'CA', ragged=True)
    pocket = {'x': dev(pb.x), 'one_hot': dev(pb.one_hot), 'size': dev(pb.size), 'mask': dev(pb.mask)}
    f_phar, f_pocket, pm, qm = ddpm.sample_given_pocket(pocket, torch.from_numpy(pb.num_nodes_phar), return_frames=4,
                                                        timesteps=8, seed=5)
    one, one_p, _, _ = ddpm.sample_given_pocket(pocket, torch.from_numpy(pb.num_nodes_phar), timesteps=8, seed=5)
    assert f_phar.shape == (4,) + tuple(one.shape) and f_pocket.shape == (4,) + tuple(one_p.shape)
    assert torch.allclose(f_phar[0], one, atol=1e-4 * max(1.0, float(one.abs().max())))
    for k in range(1, 4):
        shift = (f_pocket[k][:, :3] - pocket['x']).cpu().numpy()
        for b in range(3):
            sb = shift[pb.mask == b]
            assert np.abs(sb - sb[0]).max() < 1e-3 * max(1.0, np.abs(sb).max())           # rigid translation per sample
        assert torch.equal(f_pocket[k][:, 3:], one_p[:, 3:])


def test_sampling_from_processed_dataset(tmp_path):
    """Dataset path (NPZ schema -> collate -> size prior -> sample_given_pocket), as validation sampling uses it."""
    from argparse import Namespace
    from cmdgen_amd.dataset import ProcessedLigandPharPocketDataset, write_synthetic_npz
    from cmdgen_amd.lightning_modules import PharPocketDDPM
    from helpers import HIST
    write_synthetic_npz(str(tmp_path / 'test.npz'), n_complexes=5, seed=4)
    hp = dict(outdir='o', dataset='crossdock', datadir=str(tmp_path), batch_size=3, lr=1e-4,
              egnn_params=Namespace(device='cuda', edge_cutoff=6.0, joint_nf=32, hidden_nf=64, n_layers=2, attention=True,
                                    tanh=True, norm_constant=1, inv_sublayers=1, sin_embedding=False,
                                    aggregation_method='sum', normalization_factor=100),
              diffusion_params=Namespace(diffusion_steps=500, diffusion_noise_schedule='polynomial_2',
                                         diffusion_noise_precision=1e-5, diffusion_loss_type='l2', normalize_factors=[1, 4]),
              num_workers=0, augment_noise=0, augment_rotation=False, clip_grad=True, eval_epochs=50,
              eval_params=Namespace(n_eval_samples=7, eval_batch_size=3), mode='pocket_conditioning',
              node_histogram=HIST, pocket_representation='CA')
    model = PharPocketDDPM(**hp).cuda()
    model.setup('test')
    torch.manual_seed(0)
    out = model.sample_given_pocket_dataset(7, model.test_dataset, batch_size=3, timesteps=10)
    assert len(out) == 7                                                     # 3 + 3 + 1, cycling through 5 complexes
    for x, t, ref in out:
        assert x.shape[1] == 3 and len(x) == len(t) and 3 <= len(x) <= 25 and torch.isfinite(x).all()
        assert int(t.min()) >= 0 and int(t.max()) < 8 and ref.shape[1] == 3


def test_device_noise_is_standard_normal_and_keyed():
    """The production noise source (Philox4x32-10 + Box-Muller): moments of N(0,1), no correlation between
    neighbouring components, reproducible, and distinct per (seed, pocket id, draw)."""
    h = hip_backend.Handle(ModelConfig(hidden_nf=64, n_layers=1).as_dict(), 0)
    z = h.debug_noise(seed=123, pocket_id=7, draw=3, n_nodes=200000, width=11).cpu().numpy().astype(np.float64)
    n = z.size
    assert abs(z.mean()) < 4 / np.sqrt(n) and abs(z.var() - 1) < 4 * np.sqrt(2 / n)
    assert abs((z ** 3).mean()) < 0.02 and abs((z ** 4).mean() - 3) < 0.05
    assert np.abs(z).max() < 6.5 and np.isfinite(z).all()
    flat = z.ravel()
    assert abs(np.corrcoef(flat[:-1], flat[1:])[0, 1]) < 5 / np.sqrt(n)
    assert abs(np.corrcoef(z[:-1, 0], z[1:, 0])[0, 1]) < 5 / np.sqrt(len(z))
    frac = (np.abs(flat) < 1).mean()
    assert abs(frac - 0.682689) < 0.002
    again = h.debug_noise(123, 7, 3, 1000).cpu().numpy()
    assert np.array_equal(again, z[:1000].astype(np.float32))
    for kw in (dict(seed=124, pocket_id=7, draw=3), dict(seed=123, pocket_id=8, draw=3), dict(seed=123, pocket_id=7, draw=4)):
        other = h.debug_noise(n_nodes=1000, **kw).cpu().numpy()
        assert abs(np.corrcoef(other.ravel(), z[:1000].ravel())[0, 1]) < 0.05


def test_library_schedule_fallback_matches_host_table():
    """Without cmdgen_set_step_table the library evaluates the per-step scalars itself (libm, fp32); a chain run
    with it agrees with the host-table chain to fp32 round-off."""
    name = 'ca_h256_K5'
    cfg, sd, pb, K = chain_case(G4, name)
    h = hip_backend.Handle(cfg.as_dict(), 0)
    h.load_state_dict(sd)
    h.set_layout(pb.num_nodes_phar, pb.size)
    noise = dev(G4[name + '/noise'])
    a, _, _ = h.sample_chain(dev(pb.x), dev(pb.one_hot), K, noise=noise, use_graph=False)      # library table
    want = G4[name + '/xh_phar']
    assert rms(a[:, :3].cpu().numpy(), want[:, :3]) <= 1e-4 * max(1.0, float(np.abs(want[:, :3]).max()))
    assert np.array_equal(a[:, 3:].cpu().numpy(), want[:, 3:])


def test_c_api_demo_runs_without_python_or_torch(tmp_path):
    """examples/c_api_demo.cpp: a plain C++ host program against include/cmdgen_hip.h + libcmdgen_hip.so."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / 'c_api_demo')
    libdir = os.path.join(root, 'cmdgen_amd')
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O2', '-I' + os.path.join(root, 'include'),
           os.path.join(root, 'examples', 'c_api_demo.cpp'), '-L' + libdir, '-lcmdgen_hip', '-Wl,-rpath,' + libdir, '-o', exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    assert 'evaluations 51' in r.stdout and 'one-hot rows valid 1' in r.stdout


@pytest.mark.parametrize('K', [200, 1000])
def test_long_chain_matches_oracle(K):
    """200 strided steps and the FULL 1000-step chain of a T=1000 model with injected noise, HIP vs oracle on the same draws.
    Every evaluation's minimum distance to the 6 A cutoff is monitored on the oracle side: if a pair ever sits
    within 1e-4 A the hard-threshold graph may legitimately differ and the comparison is skipped for that seed."""
    from oracle import ref_cpu
    torch.set_num_threads(min(16, torch.get_num_threads()))
    cfg = ModelConfig(timesteps=1000)
    # coord_gain = 1: a trained-like coordinate head, so that eps_x (O(1)) really steers the chain - with the
    # reference's initial gain of 1e-3 the network's eps_x is below one ulp of the inflated coordinates
    sd = make_state_dict(cfg, seed=2, coord_gain=1.0)
    p = ref_cpu.to_torch_params(sd)
    for first in (9000, 9100, 9200, 9300, 9400):
        pb = make_pockets(2, 'CA', n_phar=9, first_index=first)
        nl = int(pb.num_nodes_phar.sum())
        noise = torch.randn((K + 2, nl, 11), generator=torch.Generator().manual_seed(first))
        margins = []
        orig = ref_cpu.get_edges

        def watched(mask, x, cutoff):
            margins.append(min_cutoff_margin(x.numpy(), mask.numpy(), cutoff))
            return orig(mask, x, cutoff)
        ref_cpu.get_edges = watched
        try:
            tape = iter(noise)
            pocket = {'x': torch.from_numpy(pb.x), 'one_hot': torch.from_numpy(pb.one_hot),
                      'size': torch.from_numpy(pb.size), 'mask': torch.from_numpy(pb.mask)}
            with torch.no_grad():
                want, want_p, _, _ = ref_cpu.sample_given_pocket(p, cfg.as_dict(), pocket, pb.num_nodes_phar, timesteps=K,
                                                                 noise=lambda shape: next(tape))
        finally:
            ref_cpu.get_edges = orig
        if min(margins) < 1e-4:
            continue
        h = handle_for(cfg, 'seed2_T1000_gain1', sd)
        h.set_layout(pb.num_nodes_phar, pb.size)
        from cmdgen_amd.equivariant_diffusion.en_diffusion import EnVariationalDiffusion  # noqa: F401
        table = ref_cpu.gamma_table(cfg.noise_schedule, cfg.timesteps, cfg.noise_precision)
        coef = ref_cpu.step_coefficients(table, cfg.timesteps, K).numpy()
        g0 = table[0]
        final = np.array([[float(torch.sqrt(torch.sigmoid(g0))), float(torch.sqrt(torch.sigmoid(-g0))),
                           float(torch.exp(0.5 * g0)), 0.0]], np.float32)
        h.set_step_table(K, np.concatenate([coef, final]))
        got, got_p, _ = h.sample_chain(dev(pb.x), dev(pb.one_hot), K, noise=noise.cuda())
        want = want.numpy()
        scale = max(1.0, float(np.abs(want[:, :3]).max()))
        err = rms(got[:, :3].cpu().numpy(), want[:, :3])
        print(f'K={K}: coordinate RMS vs oracle {err:.3e} A (max |x| {scale:.1f} A, min cutoff margin {min(margins):.2e} A)')
        assert err <= 1e-4 * scale, (first, min(margins))
        assert np.array_equal(got[:, 3:].cpu().numpy(), want[:, 3:])
        assert rms(got_p.cpu().numpy(), want_p.numpy()) <= 1e-4 * max(1.0, float(np.abs(want_p.numpy()).max()))
        return
    pytest.skip('every candidate seed came within 1e-4 A of the cutoff')


@pytest.mark.parametrize('seed', range(16))
def test_fuzz_hyperparameters_and_layouts(seed):
    """Randomised configurations outside the shipped ones: flags off, complete graph (no cutoff), no time
    conditioning, other vocabulary / joint sizes, norm constants, 1-5 blocks, ragged batches - HIP vs oracle."""
    from oracle import ref_cpu
    rng = np.random.Generator(np.random.PCG64(4242 + seed))
    cfg = ModelConfig(
        phar_nf=int(rng.integers(3, 13)), residue_nf=int(rng.integers(4, 25)), joint_nf=int(rng.choice([8, 16, 32, 48])),
        hidden_nf=int(rng.choice([64, 128, 256])), n_layers=int(rng.integers(1, 6)),
        attention=bool(rng.integers(0, 2)), tanh=bool(rng.integers(0, 2)),
        norm_constant=float(rng.choice([0.0, 1.0, 2.5])), normalization_factor=float(rng.choice([1.0, 100.0])),
        edge_cutoff=None if rng.random() < 0.25 else float(rng.choice([4.0, 6.0, 9.0])),
        condition_time=bool(rng.random() < 0.8))
    sd = make_state_dict(cfg, seed=1000 + seed, coord_gain=float(rng.choice([1e-3, 0.3])))
    B = int(rng.integers(1, 9))
    nph = rng.integers(1, 14, size=B)
    npk = rng.integers(1, 50, size=B)
    pm, qm = np.repeat(np.arange(B), nph), np.repeat(np.arange(B), npk)
    for _ in range(50):
        xq = (rng.normal(size=(len(qm), 3)) * 5.0).astype(np.float32)
        xp = (rng.normal(size=(len(pm), 3)) * 3.0).astype(np.float32)
        if cfg.edge_cutoff is None or min_cutoff_margin(np.concatenate([xp, xq]), np.concatenate([pm, qm]), cfg.edge_cutoff) > 2e-3:
            break
    else:
        pytest.skip('no layout with a safe cutoff margin found')
    xh_phar = np.concatenate([xp, rng.normal(size=(len(pm), cfg.phar_nf)).astype(np.float32)], 1)
    oh = np.eye(cfg.residue_nf, dtype=np.float32)[rng.integers(0, cfg.residue_nf, size=len(qm))] / 4.0
    xh_pocket = np.concatenate([xq, oh], 1).astype(np.float32)
    t = rng.uniform(0.0, 1.0, size=(B, 1)).astype(np.float32)
    p = ref_cpu.to_torch_params(sd)
    with torch.no_grad():
        want, want_q = ref_cpu.dynamics_forward(p, cfg.as_dict(), torch.from_numpy(xh_phar), torch.from_numpy(xh_pocket),
                                                torch.from_numpy(t), torch.from_numpy(pm), torch.from_numpy(qm))
    h = hip_backend.Handle(cfg.as_dict(), 0)
    h.load_state_dict(sd)
    h.set_layout(nph, npk)
    got, got_q = h.dynamics_forward(dev(xh_phar), dev(xh_pocket), dev(t))
    want, want_q = want.numpy(), want_q.numpy()
    tol = 5e-5 * max(1.0, float(np.abs(want).max()))       # complete graphs at normalization_factor 1 sum hundreds of terms
    assert float(np.abs(got.cpu().numpy() - want).max()) <= tol, (cfg, B)
    assert float(np.abs(got_q.cpu().numpy() - want_q).max()) <= 5e-5 * max(1.0, float(np.abs(want_q).max()))
    h.close()


def test_simple_conditional_mode_python_api():
    """mode 'pocket_conditioning_simple' (SimpleConditionalDDPM, conditional_model.py:481-525) through the Python
    mirror, against the reference's golden chain: no COM projection, pocket centred once."""
    from cmdgen_amd.equivariant_diffusion.dynamics import EGNNDynamics
    from cmdgen_amd.equivariant_diffusion.conditional_model import SimpleConditionalDDPM
    name = 'simple_h64_K5'
    cfg, sd, pb, K = chain_case(G4, name)
    dyn = EGNNDynamics(phar_nf=8, residue_nf=20, n_dims=3, joint_nf=32, hidden_nf=cfg.hidden_nf, n_layers=cfg.n_layers,
                       attention=True, tanh=True, norm_constant=1, inv_sublayers=1, normalization_factor=100,
                       aggregation_method='sum', edge_cutoff=6.0, update_pocket_coords=False)
    ddpm = SimpleConditionalDDPM(dynamics=dyn, phar_nf=8, residue_nf=20, n_dims=3, timesteps=500,
                                 noise_schedule='polynomial_2', noise_precision=1e-5, loss_type='l2', norm_values=[1, 4],
                                 size_histogram=np.ones((30, 70)))
    ddpm.load_state_dict({k[len('ddpm.'):]: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    ddpm = ddpm.cuda()
    pocket = {'x': dev(pb.x), 'one_hot': dev(pb.one_hot), 'size': dev(pb.size), 'mask': dev(pb.mask)}
    xh_phar, xh_pocket, pm, _ = ddpm.sample_given_pocket(pocket, torch.from_numpy(pb.num_nodes_phar), timesteps=K,
                                                         noise=dev(G4[name + '/noise']))
    want, wp = G4[name + '/xh_phar'], G4[name + '/xh_pocket']
    assert rms(xh_phar[:, :3].cpu().numpy(), want[:, :3]) <= 1e-4 * max(1.0, float(np.abs(want[:, :3]).max()))
    assert np.array_equal(xh_phar[:, 3:].cpu().numpy(), want[:, 3:])
    assert rms(xh_pocket.cpu().numpy(), wp) <= 1e-4 * max(1.0, float(np.abs(wp).max()))
    # the pocket is only centred, never translated by the samples
    for b in range(len(pb.size)):
        assert np.abs(xh_pocket[:, :3].cpu().numpy()[pb.mask == b].mean(0)).max() < 1e-3
    assert ddpm.subspace_dimensionality(torch.tensor([5])).item() == 15


@pytest.mark.parametrize('seed', range(8))
def test_fuzz_chains(seed):
    """Randomised sampler settings (normalisation factors, strided step counts, T, ragged layouts, with and
    without the COM projection) on injected noise: HIP chain vs oracle chain, per-step states included."""
    from oracle import ref_cpu
    rng = np.random.Generator(np.random.PCG64(777 + seed))
    T = int(rng.choice([100, 500, 1000]))
    cfg = ModelConfig(hidden_nf=int(rng.choice([64, 128])), n_layers=int(rng.integers(1, 4)), timesteps=T,
                      norm_values=(float(rng.choice([1.0, 2.0])), float(rng.choice([1.0, 4.0, 8.0]))),
                      no_com_projection=bool(rng.integers(0, 2)), phar_nf=int(rng.integers(4, 10)))
    sd = make_state_dict(cfg, seed=300 + seed, coord_gain=float(rng.choice([1e-3, 1.0])))
    K = int(rng.choice([1, 3, 7]))
    p = ref_cpu.to_torch_params(sd)
    for attempt in range(6):
        pb = make_pockets(int(rng.integers(1, 6)), 'CA', ragged=True, first_index=20000 + 100 * seed + attempt)
        nph = rng.integers(1, 12, size=len(pb.size))
        nl = int(nph.sum())
        noise = torch.randn((K + 2, nl, 3 + cfg.phar_nf), generator=torch.Generator().manual_seed(seed * 10 + attempt))
        margins = []
        orig = ref_cpu.get_edges

        def watched(mask, x, cutoff):
            margins.append(min_cutoff_margin(x.numpy(), mask.numpy(), cutoff))
            return orig(mask, x, cutoff)
        ref_cpu.get_edges = watched
        try:
            tape = iter(noise)
            pocket = {'x': torch.from_numpy(pb.x), 'one_hot': torch.from_numpy(pb.one_hot),
                      'size': torch.from_numpy(pb.size), 'mask': torch.from_numpy(pb.mask)}
            with torch.no_grad():
                want, want_p, _, _, chain = ref_cpu.sample_given_pocket(p, cfg.as_dict(), pocket, nph, timesteps=K,
                                                                       noise=lambda shape: next(tape), return_chain=True)
        finally:
            ref_cpu.get_edges = orig
        if min(margins) > 2e-3:
            break
    else:
        pytest.skip('no seed with a safe cutoff margin')
    h = hip_backend.Handle(cfg.as_dict(), 0)
    h.load_state_dict(sd)
    h.set_layout(nph, pb.size)
    got, got_p, z_steps = h.sample_chain(dev(pb.x), dev(pb.one_hot), K, noise=noise.cuda(), want_steps=True,
                                         use_graph=bool(seed & 1))          # library-side schedule table
    want = want.numpy()
    for k in range(K):
        zs = chain[k + 1].numpy()
        assert float(np.abs(z_steps[k].cpu().numpy() - zs).max()) <= 1e-4 * max(1.0, float(np.abs(zs).max())), k
    assert rms(got[:, :3].cpu().numpy(), want[:, :3]) <= 1e-4 * max(1.0, float(np.abs(want[:, :3]).max()))
    assert np.array_equal(got[:, 3:].cpu().numpy(), want[:, 3:])
    assert rms(got_p.cpu().numpy(), want_p.numpy()) <= 1e-4 * max(1.0, float(np.abs(want_p.numpy()).max()))
    h.close()
