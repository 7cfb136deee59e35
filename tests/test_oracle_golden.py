"""CPU suite: the oracle (oracle/ref_cpu.py) against the golden vectors captured
from the real reference (tests/golden/make_golden.py).  This is what pins the
oracle; the GPU parity tests then compare the HIP path with the oracle."""
import numpy as np
import pytest
import torch

from helpers import (load_golden, cases_of, dynamics_case, chain_case, NoiseTape, rms)
from oracle import ref_cpu
from cmdgen_amd.synthetic import gamma_table


def test_g1_gamma_tables_bit_exact():
    g = load_golden('g1_schedule.npz')
    for T in (100, 500, 1000):
        want = g[f'gamma_T{T}']
        got = ref_cpu.gamma_table('polynomial_2', T, 1e-5).numpy()
        assert got.dtype == np.float32 and np.array_equal(got, want)
        # the product's own table builder must agree bit for bit too
        assert np.array_equal(gamma_table('polynomial_2', T, 1e-5), want)
    # spot values quoted in SURVEY.md App. A.2
    t500 = g['gamma_T500']
    assert abs(t500[0] + 11.5129) < 1e-3 and abs(t500[500] - 11.5113) < 1e-3


def test_g1_step_coefficients_bit_exact():
    g = load_golden('g1_schedule.npz')
    table = torch.from_numpy(g['gamma_T500'])
    for K in (5, 50, 500):
        got = ref_cpu.step_coefficients(table, 500, K).numpy()
        assert np.array_equal(got, g[f'coef_T500_K{K}'])


def test_g3_edges_order_selfloops_cutoff():
    g = load_golden('g3_edges.npz')
    row, col = ref_cpu.get_edges(torch.from_numpy(g['mask']), torch.from_numpy(g['x']), 6.0)
    e = np.stack([row.numpy(), col.numpy()])
    assert np.array_equal(e, g['edges'])
    pairs = set(map(tuple, e.T))
    assert (0, 1) in pairs and (1, 0) in pairs          # distance exactly 6.0 is kept (<=)
    assert (0, 2) not in pairs                          # 6.5 is not
    assert all((i, i) in pairs for i in range(40))      # self loops (Q1)
    assert all(g['mask'][i] == g['mask'][j] for i, j in pairs)
    # row-major sorted
    key = e[0].astype(np.int64) * 1000 + e[1]
    assert np.all(np.diff(key) > 0)


@pytest.mark.parametrize('name', cases_of(load_golden('g2_dynamics.npz')))
def test_g2_dynamics_forward(name):
    g = load_golden('g2_dynamics.npz')
    cfg, sd, inp = dynamics_case(g, name)
    p = ref_cpu.to_torch_params(sd)
    trace = {}
    with torch.no_grad():
        eps_phar, eps_pocket = ref_cpu.dynamics_forward(
            p, cfg.as_dict(), torch.from_numpy(inp['xh_phar']), torch.from_numpy(inp['xh_pocket']),
            torch.from_numpy(inp['t']), torch.from_numpy(inp['mask_phar']),
            torch.from_numpy(inp['mask_pocket']), trace=trace)
    e = np.stack([trace['row'].numpy(), trace['col'].numpy()])
    assert np.array_equal(e, g[name + '/edges'])
    want = g[name + '/eps_phar']
    # same aten kernels in the same order: expect (near) bit equality
    assert np.abs(eps_phar.numpy() - want).max() <= 2e-6 * max(1.0, np.abs(want).max())
    if name + '/eps_pocket' in g:
        assert np.abs(eps_pocket.numpy() - g[name + '/eps_pocket']).max() <= 2e-6 * max(1.0, np.abs(g[name + '/eps_pocket']).max())
        nl = len(inp['mask_phar'])
        for b in range(cfg.n_layers):
            hb, xb = trace['h_block'][b].numpy(), trace['x_block'][b].numpy()
            assert np.abs(hb[:nl] - g[name + f'/block{b}_h_phar']).max() < 2e-5
            assert np.abs(hb[nl:nl + 16] - g[name + f'/block{b}_h_pocket_head']).max() < 2e-5
            assert np.abs(xb[:nl] - g[name + f'/block{b}_x_phar']).max() < 2e-5
            # pocket rows never move in conditional mode
            assert np.array_equal(xb[nl:], inp['xh_pocket'][:, :3])


@pytest.mark.parametrize('name', cases_of(load_golden('g4_chains.npz')))
def test_g4_chain_with_injected_noise(name):
    g = load_golden('g4_chains.npz')
    cfg, sd, pb, K = chain_case(g, name)
    p = ref_cpu.to_torch_params(sd)
    pocket = {'x': torch.from_numpy(pb.x), 'one_hot': torch.from_numpy(pb.one_hot),
              'size': torch.from_numpy(pb.size), 'mask': torch.from_numpy(pb.mask)}
    tape = NoiseTape(g[name + '/noise'])
    with torch.no_grad():
        xh_phar, xh_pocket, phar_mask, _, chain = ref_cpu.sample_given_pocket(
            p, cfg.as_dict(), pocket, pb.num_nodes_phar, timesteps=K, noise=tape, return_chain=True)
    assert tape.i == K + 2                       # T+2 Gaussian draws per chain
    assert np.array_equal(phar_mask.numpy(), g[name + '/phar_mask'])
    want = g[name + '/xh_phar']
    assert rms(xh_phar[:, :3].numpy(), want[:, :3]) < 1e-4 * max(1.0, np.abs(want[:, :3]).max())
    assert np.array_equal(xh_phar[:, 3:].numpy(), want[:, 3:])          # one-hot types exact
    assert rms(xh_pocket.numpy(), g[name + '/xh_pocket']) < 1e-4 * max(1.0, np.abs(g[name + '/xh_pocket']).max())
    if name + '/z_steps' in g:
        zs = g[name + '/z_steps']
        for k in range(K):
            assert np.abs(chain[k + 1].numpy() - zs[k]).max() < 1e-4 * max(1.0, np.abs(zs[k]).max())


def test_g8_node_count_prior():
    g = load_golden('g8_nodes.npz')
    got = ref_cpu.n1_given_n2_log_prob(g['hist'], g['n1'], g['n2']).numpy()
    assert np.allclose(got, g['logp'], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('mode', ['train', 'eval'])
def test_g6_loss_terms(mode):
    """ConditionalDDPM.forward's 12 loss terms with t_int and the Gaussian draws pinned (t = 0 and t = T included)."""
    from helpers import loss_case
    g = load_golden('g6_loss.npz')
    cfg, sd, phar, pocket, hist = loss_case(g)
    p = ref_cpu.to_torch_params(sd)
    with torch.no_grad():
        terms = ref_cpu.ddpm_forward(p, cfg.as_dict(), phar, pocket, torch.from_numpy(g['t_int']),
                                     [torch.from_numpy(g['eps0']), torch.from_numpy(g['eps1'])], mode == 'train', hist)
    names = ['delta_log_px', 'error_t_phar', 'error_t_pocket', 'SNR_weight', 'loss_0_x_phar', 'loss_0_x_pocket',
             'loss_0_h', 'neg_log_constants', 'kl_prior', 'log_pN', 't_int', 'xh_phar_hat']
    for n, v in zip(names, terms[:-1]):
        want = g[f'{mode}/{n}']
        got = np.asarray(v.numpy() if torch.is_tensor(v) else v, dtype=np.float32)
        assert np.allclose(got, want, rtol=2e-5, atol=2e-5 * max(1.0, float(np.abs(want).max()))), n
    assert abs(float(terms[-1]['eps_hat_phar_x']) - float(g[f'{mode}/info_eps_hat_phar_x'])) < 1e-6
    nll = ref_cpu.nll_from_terms(terms, cfg.as_dict(), phar['size'], pocket['size'], mode == 'train')
    assert nll.shape == (4,) and bool(torch.isfinite(nll).all())


# ---------------------------------------------------------------- joint model (G9)
from helpers import JointNoiseTape, joint_cfg, joint_cases, joint_inpaint_case  # noqa: E402
from cmdgen_amd.synthetic import make_state_dict  # noqa: E402

G9 = load_golden('g9_joint.npz')


def _t(d):
    return {k: torch.from_numpy(np.asarray(v).copy()) for k, v in d.items()}


@pytest.mark.parametrize('name', joint_cases(G9, 'dyn'))
def test_g9_joint_dynamics(name):
    H, L, B, R, seed, first = [int(v) for v in G9[f'dyn/{name}/meta']]
    cfg = joint_cfg(H, L, R)
    p = ref_cpu.to_torch_params(make_state_dict(cfg, seed=seed, coord_gain=1.0))
    k = f'dyn/{name}/'
    with torch.no_grad():
        ep, eq = ref_cpu.dynamics_forward(p, cfg.as_dict(), *[torch.from_numpy(G9[k + n]) for n in
                                          ('xh_phar', 'xh_pocket', 't', 'phar_mask', 'pocket_mask')])
    assert np.abs(ep.numpy() - G9[k + 'eps_phar']).max() < 2e-5
    assert np.abs(eq.numpy() - G9[k + 'eps_pocket']).max() < 2e-5
    assert np.abs(G9[k + 'eps_pocket'][:, :3]).max() > 1e-3        # pocket nodes do move in joint mode


@pytest.mark.parametrize('name', joint_cases(G9, 'sample'))
def test_g9_joint_sample(name):
    H, L, B, R, seed, K = [int(v) for v in G9[f'sample/{name}/meta']]
    cfg = joint_cfg(H, L, R)
    p = ref_cpu.to_torch_params(make_state_dict(cfg, seed=seed, coord_gain=1.0))
    k = f'sample/{name}/'
    nl, npk = G9[k + 'num_phar'], G9[k + 'num_pocket']
    tape = JointNoiseTape(G9[k + 'noise'], int(nl.sum()), int(npk.sum()))
    with torch.no_grad():
        xh_phar, xh_pocket, pm, qm, chain = ref_cpu.joint_sample(p, cfg.as_dict(), B, nl, npk, timesteps=K,
                                                                 noise=tape, return_chain=True)
    assert tape.i == K + 2 and tape.sub == 0
    for s, (zp, zq) in enumerate(chain):
        got = np.concatenate([zp.numpy().ravel(), zq.numpy().ravel()])
        assert np.abs(got - G9[k + 'z_steps'][s]).max() < 1e-4 * max(1.0, np.abs(got).max())
    assert np.array_equal(xh_phar.numpy()[:, 3:], G9[k + 'xh_phar'][:, 3:])
    assert np.array_equal(xh_pocket.numpy()[:, 3:], G9[k + 'xh_pocket'][:, 3:])
    assert np.abs(xh_phar.numpy()[:, :3] - G9[k + 'xh_phar'][:, :3]).max() < 1e-3
    assert np.abs(xh_pocket.numpy()[:, :3] - G9[k + 'xh_pocket'][:, :3]).max() < 1e-3


@pytest.mark.parametrize('name', joint_cases(G9, 'inpaint'))
def test_g9_joint_inpaint(name):
    cfg, sd, phar, pocket, K, resamplings, jump = joint_inpaint_case(G9, name)
    p = ref_cpu.to_torch_params(sd)
    k = f'inpaint/{name}/'
    tape = JointNoiseTape(G9[k + 'noise'], len(phar['mask']), len(pocket['mask']))
    with torch.no_grad():
        xh_phar, xh_pocket, pm, qm = ref_cpu.joint_inpaint(
            p, cfg.as_dict(), _t(phar), _t(pocket), torch.from_numpy(G9[k + 'phar_fixed']),
            torch.from_numpy(G9[k + 'pocket_fixed']), resamplings=resamplings, jump_length=jump,
            timesteps=K, noise=tape)
    assert tape.i == len(G9[k + 'noise']) and tape.sub == 0
    assert np.array_equal(xh_phar.numpy()[:, 3:], G9[k + 'xh_phar'][:, 3:])
    assert np.array_equal(xh_pocket.numpy()[:, 3:], G9[k + 'xh_pocket'][:, 3:])
    assert np.abs(xh_phar.numpy()[:, :3] - G9[k + 'xh_phar'][:, :3]).max() < 1e-3
    assert np.abs(xh_pocket.numpy()[:, :3] - G9[k + 'xh_pocket'][:, :3]).max() < 1e-3


def test_g9_repaint_schedules():
    n = 0
    for key, want in G9.items():
        if key.startswith('schedule/'):
            r, j, T = [int(s[1:]) for s in key.split('/')[1].split('_')]
            assert ref_cpu.get_repaint_schedule(r, j, T) == want.tolist()
            n += 1
    assert n == 7


@pytest.mark.parametrize('mode', ['train', 'eval'])
def test_g9_joint_loss_terms(mode):
    from helpers import joint_loss_case, LOSS_NAMES
    cfg, sd, phar, pocket, hist = joint_loss_case(G9)
    p = ref_cpu.to_torch_params(sd)
    tape = JointNoiseTape(G9[f'loss/{mode}/noise'], len(phar['mask']), len(pocket['mask']))
    with torch.no_grad():
        terms = ref_cpu.joint_ddpm_forward(p, cfg.as_dict(), phar, pocket, torch.from_numpy(G9['loss/t_int']), tape,
                                           training=(mode == 'train'), histogram=hist)
    assert tape.i == (1 if mode == 'train' else 2)
    for n, v in zip(LOSS_NAMES, terms[:-1]):
        want = G9[f'loss/{mode}/{n}']
        got = np.asarray(v.numpy() if torch.is_tensor(v) else v, dtype=np.float32)
        assert got.shape == want.shape, n
        assert np.abs(got - want).max() <= 2e-5 * max(1.0, np.abs(want).max()), (n, got, want)
    for kk, v in terms[-1].items():
        assert abs(float(v) - float(G9[f'loss/{mode}/info_{kk}'])) < 1e-5, kk
    assert np.abs(G9[f'loss/{mode}/error_t_pocket']).max() > 0        # the joint loss has pocket terms


# ---------------------------------------------------------------- training step (G11)
def _oracle_training_grads():
    from helpers import loss_case
    g6 = load_golden('g6_loss.npz')
    cfg, sd, phar, pocket, hist = loss_case(g6)
    p = ref_cpu.to_torch_params(sd)
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items() if k.startswith('dynamics.')}
    p2 = dict(p); p2.update(leaves)
    terms = ref_cpu.ddpm_forward(p2, cfg.as_dict(), phar, pocket, torch.from_numpy(g6['t_int']), [torch.from_numpy(g6['eps0'])],
                                 training=True, histogram=hist)
    nll = ref_cpu.nll_from_terms(terms, cfg.as_dict(), phar['size'], pocket['size'], training=True)
    loss = nll.mean(0)
    loss.backward()
    return loss.detach(), nll.detach(), leaves


def test_g11_oracle_autograd_matches_reference_gradients():
    """The oracle is differentiable torch code: its autograd gradients of the training loss must equal the
    reference's (make_golden_grad.py) - this pins the oracle as the checker of the HIP backward pass."""
    g = load_golden('g11_train.npz')
    loss, nll, leaves = _oracle_training_grads()
    assert abs(float(loss) - float(g['step0/loss'])) < 1e-6
    assert np.abs(nll.numpy() - g['step0/nll']).max() < 1e-5
    n = 0
    for key, want in g.items():
        if key.startswith('grad/'):
            name = key[len('grad/'):]
            if name == 'gamma.gamma':               # requires_grad=False in the reference (en_diffusion.py:1180-1182)
                assert not want.any()
                continue
            got = leaves[name].grad
            got = np.zeros_like(want) if got is None else got.numpy()
            assert np.abs(got - want).max() <= 2e-5 * max(np.abs(want).max(), 1e-3), name
            n += 1
    assert n == 50


@pytest.mark.parametrize('case', ['s2_sum', 's1_mean', 's3_mean'])
def test_g19_oracle_autograd_matches_reference_gradients_with_egnn_options(case):
    """inv_sublayers > 1 (egnn_new.py:127-131) and aggregation 'mean' (egnn_new.py:285-292) in the TRAINING loss: the oracle's autograd
    gradients against the real reference's (tests/golden/make_golden_r5.py) - the checker of the HIP backward pass with these options."""
    import dataclasses
    from helpers import loss_case
    g = load_golden('g19_train_options.npz')
    g6 = load_golden('g6_loss.npz')
    S, mean = [int(v) for v in g[f'{case}/options']]
    cfg, _, phar, pocket, hist = loss_case(g6)
    cfg = dataclasses.replace(cfg, inv_sublayers=S, aggregation_method='mean' if mean else 'sum')
    H, L, B, R, seed, first = [int(v) for v in g6['meta']]
    sd = make_state_dict(cfg, seed=seed, coord_gain=1.0)
    p = ref_cpu.to_torch_params(sd)
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items() if k.startswith('dynamics.')}
    p2 = dict(p); p2.update(leaves)
    terms = ref_cpu.ddpm_forward(p2, cfg.as_dict(), phar, pocket, torch.from_numpy(g6['t_int']), [torch.from_numpy(g6['eps0'])],
                                 training=True, histogram=hist)
    nll = ref_cpu.nll_from_terms(terms, cfg.as_dict(), phar['size'], pocket['size'], training=True)
    loss = nll.mean(0)
    loss.backward()
    assert abs(float(loss) - float(g[f'{case}/loss'])) < 2e-6
    assert np.abs(nll.detach().numpy() - g[f'{case}/nll']).max() < 1e-5
    n = 0
    for key, want in g.items():
        if key.startswith(f'{case}/grad/'):
            name = key[len(f'{case}/grad/'):]
            if name == 'gamma.gamma':
                assert not want.any()
                continue
            got = leaves[name].grad
            got = np.zeros_like(want) if got is None else got.numpy()
            assert np.abs(got - want).max() <= 2e-5 * max(np.abs(want).max(), 1e-3), name
            n += 1
    assert n == 20 + S * 10 * L + 5 * L          # 20 encoder / decoder / embedding tensors, 10 per GCL (with attention), 5 per EquivariantUpdate
