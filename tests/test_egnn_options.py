"""The two EGNN constructor options no shipped config uses but the reference implements - `inv_sublayers` > 1 (several GCLs per
EquivariantBlock, egnn_new.py:127-131, :152-154) and `aggregation_method='mean'` (egnn_new.py:288-292) - against golden G16, captured from
the real reference by tests/golden/make_golden_r4.py g16: the CPU oracle (not gpu) and the HIP path through every node / edge kernel
family a layout can select (gpu)."""
import numpy as np
import pytest
import torch

from helpers import load_golden, masks_from_sizes, rms, NoiseTape
from cmdgen_amd.synthetic import ModelConfig, make_state_dict, make_pockets

G16 = load_golden('g16_egnn_options.npz')
DYN = sorted({k.split('/')[0] for k in G16 if not k.startswith('chain_')})
CHAINS = sorted({k.split('/')[0] for k in G16 if k.startswith('chain_')})


def dyn_case(name):
    H, L, S, mean, B, R, seed, first, ragged = [int(v) for v in G16[name + '/meta']]
    cfg = ModelConfig(hidden_nf=H, n_layers=L, residue_nf=R, timesteps=500, inv_sublayers=S, aggregation_method='mean' if mean else 'sum')
    return cfg, make_state_dict(cfg, seed=seed, coord_gain=1.0)


def chain_case(name):
    H, L, S, mean, B, R, seed, first, K = [int(v) for v in G16[name + '/meta']]
    cfg = ModelConfig(hidden_nf=H, n_layers=L, residue_nf=R, timesteps=500, inv_sublayers=S, aggregation_method='mean' if mean else 'sum',
                      noise_precision=0.05, norm_values=(1.0, 0.5))
    return cfg, make_state_dict(cfg, seed=seed, coord_gain=1e-3), make_pockets(B, 'CA', ragged=True, n_phar=8, first_index=first), K


# ----------------------------------------------------------------------------- the oracle (CPU)
@pytest.mark.parametrize('name', DYN)
def test_oracle_dynamics_forward_g16(name):
    from oracle import ref_cpu
    cfg, sd = dyn_case(name)
    pm, qm = masks_from_sizes(G16[name + '/pocket_size'], G16[name + '/num_nodes_phar'])
    with torch.no_grad():
        ep, eq = ref_cpu.dynamics_forward(ref_cpu.to_torch_params(sd), cfg.as_dict(), torch.from_numpy(G16[name + '/xh_phar']),
                                          torch.from_numpy(G16[name + '/xh_pocket']), torch.from_numpy(G16[name + '/t']),
                                          torch.from_numpy(pm), torch.from_numpy(qm))
    for got, key in ((ep, 'eps_phar'), (eq, 'eps_pocket')):
        want = G16[name + '/' + key]
        assert np.abs(got.numpy() - want).max() <= 2e-6 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize('name', CHAINS)
def test_oracle_chain_g16(name):
    from oracle import ref_cpu
    cfg, sd, pb, K = chain_case(name)
    pocket = {'x': torch.from_numpy(pb.x), 'one_hot': torch.from_numpy(pb.one_hot), 'size': torch.from_numpy(pb.size), 'mask': torch.from_numpy(pb.mask)}
    tape = NoiseTape(G16[name + '/noise'])
    with torch.no_grad():
        out = ref_cpu.sample_given_pocket(ref_cpu.to_torch_params(sd), cfg.as_dict(), pocket, pb.num_nodes_phar, timesteps=K, noise=tape)
    want = G16[name + '/xh_phar']
    assert tape.i == K + 2
    assert rms(out[0][:, :3].numpy(), want[:, :3]) < 1e-4 and np.array_equal(out[0][:, 3:].numpy(), want[:, 3:])


# ----------------------------------------------------------------------------- the HIP path
def _gpu():
    from cmdgen_amd import hip_backend
    return hip_backend, torch.device('cuda')


OPTION_SETS = [{}, {'node_mt': 16, 'node16w': 0, 'edge_mt': 16, 'coord_mt': 16}, {'node_mt': 32, 'edge_mt': 64, 'coord_mt': 64},
               {'node_mt': 64, 'node64': 0, 'edge_mt': 32, 'coord_mt': 32, 'edge_fullk': 0}, {'node64': 1, 'edge_mt': 128, 'coord_mt': 128},
               {'node64': 32}, {'node64': 8}, {'node64': 2}]


@pytest.mark.gpu
@pytest.mark.parametrize('engine', ['split', 'fp32'])
@pytest.mark.parametrize('opts', range(len(OPTION_SETS)))
@pytest.mark.parametrize('name', DYN)
def test_hip_dynamics_forward_g16(name, opts, engine):
    hip_backend, dev = _gpu()
    cfg, sd = dyn_case(name)
    o = OPTION_SETS[opts]
    if cfg.hidden_nf != 256 and any(k in o for k in ('node64', 'node16w')) and o.get('node64', 0):
        pytest.skip('plane node tiles: hidden_nf 256 only')
    if engine == 'fp32' and opts not in (0, 2):
        pytest.skip('fp32 instruction: the default layout and one forced tile set')
    h = hip_backend.Handle(cfg.as_dict(), 0)
    h.load_state_dict(sd)
    h.set_gemm_mode(engine == 'split')
    for k, v in o.items():
        if cfg.hidden_nf != 256 and (k in ('node64', 'node16w') or v == 128):
            continue
        h.set_option(k, v)
    h.set_layout(G16[name + '/num_nodes_phar'], G16[name + '/pocket_size'])
    ep, eq = h.dynamics_forward(torch.from_numpy(G16[name + '/xh_phar']).to(dev), torch.from_numpy(G16[name + '/xh_pocket']).to(dev),
                                torch.from_numpy(G16[name + '/t']).to(dev), want_pocket=True)
    want = G16[name + '/eps_phar']
    assert float(np.abs(ep.cpu().numpy() - want).max()) <= 2e-5 * max(1.0, float(np.abs(want).max()))
    wq = G16[name + '/eps_pocket']
    assert float(np.abs(eq.cpu().numpy() - wq).max()) <= 2e-5 * max(1.0, float(np.abs(wq).max()))
    h.close()


@pytest.mark.gpu
@pytest.mark.parametrize('use_graph', [False, True])
@pytest.mark.parametrize('name', CHAINS)
def test_hip_chain_g16(name, use_graph):
    hip_backend, dev = _gpu()
    from test_hip_parity_r2 import host_step_table
    cfg, sd, pb, K = chain_case(name)
    h = hip_backend.Handle(cfg.as_dict(), 0)
    h.load_state_dict(sd)
    h.set_layout(pb.num_nodes_phar, pb.size)
    h.set_step_table(K, host_step_table(cfg, K))
    x, xp, _ = h.sample_chain(torch.from_numpy(pb.x).to(dev), torch.from_numpy(pb.one_hot).to(dev), K,
                              noise=torch.from_numpy(G16[name + '/noise']).to(dev), use_graph=use_graph)
    st = h.chain_status()
    want = G16[name + '/xh_phar']
    assert rms(x[:, :3].cpu().numpy(), want[:, :3]) <= 1e-4
    assert np.array_equal(x[:, 3:].cpu().numpy(), want[:, 3:])
    assert rms(xp.cpu().numpy(), G16[name + '/xh_pocket']) <= 1e-4
    assert st['nan_resets'] == 0 and st['max_rel_com_error'] < 1e-2
    h.close()


@pytest.mark.gpu
def test_training_refuses_the_options_it_does_not_cover():
    """inv_sublayers > 1 and aggregation 'mean' train since round 5 (tests/test_hip_train.py, G19); sin_embedding (and a learned schedule,
    hidden_nf 512) still sample only - and say so."""
    import dataclasses
    hip_backend, dev = _gpu()
    cfg, _ = dyn_case('ca_h64_s2_sum')
    cfg = dataclasses.replace(cfg, sin_embedding=True)
    h = hip_backend.Handle(cfg.as_dict(), 0)
    h.load_state_dict(make_state_dict(cfg, seed=3, coord_gain=1.0))
    h.set_layout(G16['ca_h64_s2_sum/num_nodes_phar'], G16['ca_h64_s2_sum/pocket_size'])
    with pytest.raises(hip_backend.CmdgenError, match='sin_embedding False'):
        z = torch.zeros(8, device=dev)
        h._check(h.lib.cmdgen_train_forward(h.h, z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), None, None), 'cmdgen_train_forward')
    h.close()


# ----------------------------------------------------------------------------- noise_schedule='learned' (GammaNetwork, en_diffusion.py:1058-1096): sampling
# The network's normalisation (gamma_tilde(t) - gamma_tilde(0)) / (gamma_tilde(1) - gamma_tilde(0)) cancels the ~1e-5-relative rounding of its
# 1024-term sums into ~5e-4 of gamma: the reference's own values are reproducible across hosts (BLAS, thread count) only that far.  So the
# fixture stores gamma at the chain's time points, and the chain is pinned ON THOSE VALUES (tight); the network itself and the mirror's
# end-to-end plumbing are checked at the tolerance the network's conditioning allows.
G17 = load_golden('g17_learned_schedule.npz')
LEARNED = sorted({k.split('/')[0] for k in G17})


def learned_case(name):
    H, L, B, seed, first, K = [int(v) for v in G17[name + '/meta']]
    cfg = ModelConfig(hidden_nf=H, n_layers=L, residue_nf=20, timesteps=500, noise_schedule='learned')
    return cfg, make_state_dict(cfg, seed=seed, coord_gain=1e-3), make_pockets(B, 'CA', ragged=True, n_phar=8, first_index=first), K


def pinned_schedule(name):
    """The same model with the reference's gamma values at t = i / K as a K-step lookup table (round(t * K) = i: the sampler reads exactly them)."""
    cfg, sd, pb, K = learned_case(name)
    cfg_k = ModelConfig(hidden_nf=cfg.hidden_nf, n_layers=cfg.n_layers, residue_nf=20, timesteps=K)
    sd_k = {k: v for k, v in sd.items() if not k.startswith('ddpm.gamma.')}
    sd_k['ddpm.gamma.gamma'] = G17[name + '/gamma_steps'].astype(np.float32)
    return cfg_k, sd_k, pb, K


@pytest.mark.parametrize('name', LEARNED)
def test_oracle_learned_schedule_g17(name):
    from oracle import ref_cpu
    cfg, sd, pb, K = learned_case(name)
    p = ref_cpu.to_torch_params(sd)
    with torch.no_grad():
        grid = ref_cpu.gamma_lookup(ref_cpu.gamma_source(p), torch.linspace(0, 1, 101).view(-1, 1), cfg.timesteps).view(-1).numpy()
    assert np.abs(grid - G17[name + '/gamma_grid']).max() <= 2e-3 and np.all(np.diff(grid) > 0)      # monotone, [-5, 10]
    cfg_k, sd_k, pb, K = pinned_schedule(name)
    pocket = {'x': torch.from_numpy(pb.x), 'one_hot': torch.from_numpy(pb.one_hot), 'size': torch.from_numpy(pb.size), 'mask': torch.from_numpy(pb.mask)}
    tape = NoiseTape(G17[name + '/noise'])
    with torch.no_grad():
        out = ref_cpu.sample_given_pocket(ref_cpu.to_torch_params(sd_k), cfg_k.as_dict(), pocket, pb.num_nodes_phar, timesteps=K, noise=tape)
    want = G17[name + '/xh_phar']
    assert rms(out[0][:, :3].numpy(), want[:, :3]) <= 5e-6 * float(np.abs(want[:, :3]).max())      # |x| ~ 450 A (untrained weights, 1 / alpha_T = 150)
    assert np.array_equal(out[0][:, 3:].numpy(), want[:, 3:])


@pytest.mark.gpu
@pytest.mark.parametrize('use_graph', [False, True])
@pytest.mark.parametrize('name', LEARNED)
def test_hip_chain_on_the_learned_schedules_values_g17(name, use_graph):
    hip_backend, dev = _gpu()
    from oracle import ref_cpu
    cfg_k, sd_k, pb, K = pinned_schedule(name)
    h = hip_backend.Handle(cfg_k.as_dict(), 0)
    h.load_state_dict(sd_k)
    h.set_layout(pb.num_nodes_phar, pb.size)
    table = torch.from_numpy(sd_k['ddpm.gamma.gamma'])
    g0 = table[0]
    coef = np.concatenate([ref_cpu.step_coefficients(table, K, K).numpy(),
                           np.array([[float(torch.sqrt(torch.sigmoid(g0))), float(torch.sqrt(torch.sigmoid(-g0))), float(torch.exp(0.5 * g0)), 0.0]], np.float32)])
    h.set_step_table(K, coef)
    x, xp, _ = h.sample_chain(torch.from_numpy(pb.x).to(dev), torch.from_numpy(pb.one_hot).to(dev), K,
                              noise=torch.from_numpy(G17[name + '/noise']).to(dev), use_graph=use_graph)
    want = G17[name + '/xh_phar']
    assert rms(x[:, :3].cpu().numpy(), want[:, :3]) <= 5e-6 * float(np.abs(want[:, :3]).max())
    assert np.array_equal(x[:, 3:].cpu().numpy(), want[:, 3:])
    assert rms(xp.cpu().numpy(), G17[name + '/xh_pocket']) <= 5e-6 * float(np.abs(G17[name + '/xh_pocket']).max())
    h.close()


@pytest.mark.gpu
@pytest.mark.parametrize('name', LEARNED)
def test_hip_learned_schedule_through_the_python_mirror_g17(name):
    """The Python mirror end to end: ConditionalDDPM(noise_schedule='learned') loads the reference's state_dict (gamma.l1 / l2 / l3 /
    gamma_0 / gamma_1), builds the per-step scalars from the network with the reference's op sequence on the host and runs the chain on the
    device; tolerance = what the network's conditioning leaves of the reference's own values on another host (see above)."""
    from cmdgen_amd.equivariant_diffusion.dynamics import EGNNDynamics
    from cmdgen_amd.equivariant_diffusion.conditional_model import ConditionalDDPM
    cfg, sd, pb, K = learned_case(name)
    dyn = EGNNDynamics(phar_nf=8, residue_nf=20, n_dims=3, joint_nf=32, hidden_nf=cfg.hidden_nf, n_layers=cfg.n_layers, attention=True,
                       tanh=True, norm_constant=1, inv_sublayers=1, normalization_factor=100, aggregation_method='sum',
                       edge_cutoff=6.0, update_pocket_coords=False)
    ddpm = ConditionalDDPM(dynamics=dyn, phar_nf=8, residue_nf=20, n_dims=3, timesteps=500, noise_schedule='learned',
                           noise_precision=1e-5, loss_type='vlb', norm_values=[1, 4], size_histogram=np.ones((30, 70)))
    ddpm.load_state_dict({k[len('ddpm.'):]: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    ddpm = ddpm.cuda()
    with torch.no_grad():
        grid = ddpm.gamma.cpu_copy()(torch.linspace(0, 1, 101).view(-1, 1)).view(-1).numpy()
    assert np.abs(grid - G17[name + '/gamma_grid']).max() <= 2e-3
    pocket = {'x': torch.from_numpy(pb.x).cuda(), 'one_hot': torch.from_numpy(pb.one_hot).cuda(), 'size': torch.from_numpy(pb.size).cuda(),
              'mask': torch.from_numpy(pb.mask).cuda()}
    out = ddpm.sample_given_pocket(pocket, torch.from_numpy(pb.num_nodes_phar), timesteps=K, noise=torch.from_numpy(G17[name + '/noise']))
    want = G17[name + '/xh_phar']
    got = out[0].cpu().numpy()
    assert rms(got[:, :3], want[:, :3]) <= 2e-3 * float(np.abs(want[:, :3]).max())
    assert float(np.mean(np.all(got[:, 3:] == want[:, 3:], axis=1))) >= 0.9
    with pytest.raises(NotImplementedError, match='learned'):
        from cmdgen_amd.training import HipTrainer
        class _M:                      # the trainer refuses a learned schedule before it touches anything else
            mode, loss_type = 'pocket_conditioning', 'vlb'
        m = _M(); m.ddpm = ddpm
        HipTrainer(m)


# ----------------------------------------------------------------------------- sin_embedding=True (SinusoidsEmbeddingNew, egnn_new.py:174-176, :249-260)
G18 = load_golden('g18_sin_embedding.npz')
SIN_DYN = sorted({k.split('/')[0] for k in G18 if not k.startswith('chain_')})


def sin_case(name):
    H, L, S, mean, B, R, seed, first = [int(v) for v in G18[name + '/meta']]
    cfg = ModelConfig(hidden_nf=H, n_layers=L, residue_nf=R, timesteps=500, inv_sublayers=S, aggregation_method='mean' if mean else 'sum', sin_embedding=True)
    return cfg, make_state_dict(cfg, seed=seed, coord_gain=1.0)


def sin_chain_case():
    H, L, B, seed, first, K = [int(v) for v in G18['chain_h64_K20/meta']]
    cfg = ModelConfig(hidden_nf=H, n_layers=L, residue_nf=20, timesteps=500, sin_embedding=True, noise_precision=0.05, norm_values=(1.0, 0.5))
    return cfg, make_state_dict(cfg, seed=seed, coord_gain=1e-3), make_pockets(B, 'CA', ragged=True, n_phar=8, first_index=first), K


@pytest.mark.parametrize('name', SIN_DYN)
def test_oracle_sin_embedding_g18(name):
    from oracle import ref_cpu
    cfg, sd = sin_case(name)
    pm, qm = masks_from_sizes(G18[name + '/pocket_size'], G18[name + '/num_nodes_phar'])
    with torch.no_grad():
        ep, eq = ref_cpu.dynamics_forward(ref_cpu.to_torch_params(sd), cfg.as_dict(), torch.from_numpy(G18[name + '/xh_phar']),
                                          torch.from_numpy(G18[name + '/xh_pocket']), torch.from_numpy(G18[name + '/t']), torch.from_numpy(pm), torch.from_numpy(qm))
    for got, key in ((ep, 'eps_phar'), (eq, 'eps_pocket')):
        want = G18[name + '/' + key]
        assert np.abs(got.numpy() - want).max() <= 2e-6 * max(1.0, np.abs(want).max())


def test_oracle_sin_embedding_chain_g18():
    from oracle import ref_cpu
    cfg, sd, pb, K = sin_chain_case()
    pocket = {'x': torch.from_numpy(pb.x), 'one_hot': torch.from_numpy(pb.one_hot), 'size': torch.from_numpy(pb.size), 'mask': torch.from_numpy(pb.mask)}
    with torch.no_grad():
        out = ref_cpu.sample_given_pocket(ref_cpu.to_torch_params(sd), cfg.as_dict(), pocket, pb.num_nodes_phar, timesteps=K, noise=NoiseTape(G18['chain_h64_K20/noise']))
    want = G18['chain_h64_K20/xh_phar']
    assert rms(out[0][:, :3].numpy(), want[:, :3]) < 1e-4 and np.array_equal(out[0][:, 3:].numpy(), want[:, 3:])


@pytest.mark.gpu
@pytest.mark.parametrize('mt', [None, 16, 32, 64])
@pytest.mark.parametrize('name', SIN_DYN)
def test_hip_sin_embedding_g18(name, mt):
    hip_backend, dev = _gpu()
    cfg, sd = sin_case(name)
    h = hip_backend.Handle(cfg.as_dict(), 0)
    h.load_state_dict(sd)
    if mt is not None:
        for k in ('node_mt', 'edge_mt', 'coord_mt'):
            h.set_option(k, mt)
    h.set_layout(G18[name + '/num_nodes_phar'], G18[name + '/pocket_size'])
    assert h.query('gemm_split') == 0                      # the fp32 matrix instruction (the split engine's tile builders carry two scalar features)
    with pytest.raises(hip_backend.CmdgenError, match='fp32 matrix instruction'):
        h.set_gemm_mode(True)
    ep, eq = h.dynamics_forward(torch.from_numpy(G18[name + '/xh_phar']).to(dev), torch.from_numpy(G18[name + '/xh_pocket']).to(dev),
                                torch.from_numpy(G18[name + '/t']).to(dev), want_pocket=True)
    # The highest sinusoid multiplies the distance by 2 pi 1024 / 15 = 429 rad / A: coordinate noise of 1e-6 A in a later block moves its argument by
    # 4e-4 rad.  With 'sum' aggregation the coordinate updates are divided by 100 and the evaluation keeps its usual fp32 floor (the oracle in
    # float64 against itself in float32: 1.7e-6 / 2.2e-6 on ca_h64 / ca_h256); with 'mean' (divided by ~10) the floor of ca_h128_s2_mean is 8.0e-5
    # (same measurement) - the reference's own fp32 result is only defined that far.
    tol = 2e-4 if name == 'ca_h128_s2_mean' else 2e-5
    for got, key in ((ep, 'eps_phar'), (eq, 'eps_pocket')):
        want = G18[name + '/' + key]
        assert float(np.abs(got.cpu().numpy() - want).max()) <= tol * max(1.0, float(np.abs(want).max()))
    h.close()


@pytest.mark.gpu
@pytest.mark.parametrize('use_graph', [False, True])
def test_hip_sin_embedding_chain_g18(use_graph):
    hip_backend, dev = _gpu()
    from test_hip_parity_r2 import host_step_table
    cfg, sd, pb, K = sin_chain_case()
    h = hip_backend.Handle(cfg.as_dict(), 0)
    h.load_state_dict(sd)
    h.set_layout(pb.num_nodes_phar, pb.size)
    h.set_step_table(K, host_step_table(cfg, K))
    x, xp, _ = h.sample_chain(torch.from_numpy(pb.x).to(dev), torch.from_numpy(pb.one_hot).to(dev), K,
                              noise=torch.from_numpy(G18['chain_h64_K20/noise']).to(dev), use_graph=use_graph)
    want = G18['chain_h64_K20/xh_phar']
    assert rms(x[:, :3].cpu().numpy(), want[:, :3]) <= 1e-4 and np.array_equal(x[:, 3:].cpu().numpy(), want[:, 3:])
    assert rms(xp.cpu().numpy(), G18['chain_h64_K20/xh_pocket']) <= 1e-4
    h.close()
