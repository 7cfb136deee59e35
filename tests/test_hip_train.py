"""GPU parity tests of the TRAINING step: the activation-saving forward, the backward pass
(parameter gradients), AdamW(amsgrad) and the adaptive clipping, against
(a) G11 = gradients / optimizer trajectory of the real reference (tests/golden/make_golden_grad.py) and
(b) autograd through the oracle on seeded inputs.

Tolerances: forward as the sampler's evaluation (2e-5 relative); a parameter gradient tensor
max|dg| <= 2e-4 * max|g| of that tensor (fp32 sums over ~10^4 edges in a different order, float atomics).
"""
import numpy as np
import pytest
import torch

from helpers import load_golden, loss_case, HIST
from oracle import ref_cpu
from cmdgen_amd import hip_backend
from cmdgen_amd.synthetic import ModelConfig, make_state_dict, make_pockets, min_cutoff_margin

pytestmark = pytest.mark.gpu
GRAD_TOL = 2e-4


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def make_handle(cfg):
    return hip_backend.Handle(cfg.as_dict(), 0)


def flat_theta(h, sd):
    theta = torch.zeros(h.param_count(), dtype=torch.float32)
    for k, v in sd.items():
        name = k[len('ddpm.dynamics.'):] if k.startswith('ddpm.dynamics.') else None
        if name is None:
            continue
        off, cnt = h.param_offset(name)
        assert cnt == v.size
        theta[off:off + cnt] = torch.from_numpy(v.reshape(-1))
    return theta.cuda()


# ------------------------------------------------------------------ the GEMM
@pytest.mark.parametrize('ta,tb', [(False, True), (False, False), (True, False), (True, True)])
def test_training_gemm_all_layouts(ta, tb):
    h = make_handle(ModelConfig(hidden_nf=64, n_layers=1))
    g = torch.Generator().manual_seed(1)
    for M, N, K in [(64, 64, 16), (70, 33, 20), (1, 256, 514), (300, 8, 11), (257, 129, 1000), (5, 5, 3)]:
        A = torch.randn((K, M) if ta else (M, K), generator=g).cuda()
        B = torch.randn((N, K) if tb else (K, N), generator=g).cuda()
        bias = torch.randn(N, generator=g).cuda()
        want = (A.t() if ta else A).double() @ (B.t() if tb else B).double() + bias.double()
        got = h.debug_sgemm(A, B, ta=ta, tb=tb, bias=bias)
        assert (got.double() - want).abs().max() <= 1e-5 * max(1.0, float(want.abs().max())), (M, N, K)
    # sub-blocks with leading dimensions, accumulate, split-K with atomics
    A = torch.randn(5000, 40, generator=g).cuda()
    B = torch.randn(5000, 70, generator=g).cuda()
    C0 = torch.randn(24, 64, generator=g).cuda()
    want = C0.double() + A[:, 3:27].double().t() @ B[:, 2:66].double()
    C = C0.clone()
    h.debug_sgemm(A[:, 3:27], B[:, 2:66], ta=True, tb=False, C_out=C, accumulate=True, split_k=7)
    assert (C.double() - want).abs().max() <= 2e-5 * float(want.abs().max())
    C = C0.clone()
    h.debug_sgemm(A[:, 3:27], B[:, 2:66], ta=True, tb=False, C_out=C, accumulate=True, split_k=1)
    assert (C.double() - want).abs().max() <= 2e-5 * float(want.abs().max())


# ------------------------------------------------------------------ forward
def case_inputs(cfg, B, first, rng, spread=3.0):
    while True:
        pb = make_pockets(B, 'CA', ragged=True, first_index=first)
        nl = pb.num_nodes_phar
        pm = np.repeat(np.arange(B), nl)
        com = np.stack([pb.x[pb.mask == b].mean(0) for b in range(B)])
        xp = (com[pm] + rng.normal(size=(len(pm), 3)) * spread).astype(np.float32)
        if min_cutoff_margin(np.concatenate([xp, pb.x]), np.concatenate([pm, pb.mask]), 6.0) > 2e-3:
            break
        first += 13
    xh_phar = np.concatenate([xp, rng.normal(size=(len(pm), cfg.phar_nf)).astype(np.float32)], 1)
    xh_pocket = np.concatenate([pb.x, pb.one_hot / cfg.norm_values[1]], 1).astype(np.float32)
    t = rng.uniform(size=(B, 1)).astype(np.float32)
    return pb, pm, xh_phar, xh_pocket, t


@pytest.mark.parametrize('H,L,B,kw', [(64, 2, 3, {}), (256, 5, 4, {}), (128, 3, 2, {'attention': False, 'tanh': False}),
                                      (64, 1, 2, {'condition_time': False, 'edge_cutoff': None})])
def test_training_forward_and_backward_vs_oracle_autograd(H, L, B, kw):
    cfg = ModelConfig(hidden_nf=H, n_layers=L, **kw)
    sd = make_state_dict(cfg, seed=300 + H + L, coord_gain=1.0)
    h = make_handle(cfg)
    theta = flat_theta(h, sd)
    rng = np.random.Generator(np.random.PCG64(H * 10 + L))
    pb, pm, xh_phar, xh_pocket, t = case_inputs(cfg, B, 660000 + H + L, rng)
    h.set_layout(pb.num_nodes_phar, pb.size)
    eps = h.train_forward(theta, dev(xh_phar), dev(xh_pocket), dev(t))
    # oracle with autograd
    p = ref_cpu.to_torch_params(sd)
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items() if k.startswith('dynamics.')}
    p2 = dict(p); p2.update(leaves)
    want, _ = ref_cpu.dynamics_forward(p2, cfg.as_dict(), torch.from_numpy(xh_phar), torch.from_numpy(xh_pocket),
                                       torch.from_numpy(t), torch.from_numpy(pm), torch.from_numpy(pb.mask))
    got = eps.cpu().numpy()
    assert np.abs(got - want.detach().numpy()).max() <= 2e-5 * max(1.0, float(want.detach().abs().max()))
    # the inference kernels agree with the training forward as well (same weights, packed vs flat)
    h.load_state_dict(sd)
    inf, _ = h.dynamics_forward(dev(xh_phar), dev(xh_pocket), dev(t))
    assert np.abs(inf.cpu().numpy() - got).max() <= 2e-5 * max(1.0, float(np.abs(got).max()))
    # backward: a random cotangent
    d_eps = rng.normal(size=got.shape).astype(np.float32)
    (want * torch.from_numpy(d_eps)).sum().backward()
    grad = torch.zeros_like(theta)
    h.train_backward(dev(d_eps), grad)
    grad = grad.cpu().numpy()
    checked = 0
    for name, leaf in leaves.items():
        nm = name[len('dynamics.'):]
        off, cnt = h.param_offset(nm)
        g_want = np.zeros(cnt, np.float32) if leaf.grad is None else leaf.grad.numpy().reshape(-1)
        g_got = grad[off:off + cnt]
        scale = max(float(np.abs(g_want).max()), 1e-6)
        assert np.abs(g_got - g_want).max() <= GRAD_TOL * scale, (nm, float(np.abs(g_got - g_want).max()), scale)
        checked += cnt
    assert checked <= grad.size < checked + 4 * len(leaves) + 4        # tensors are 16-byte aligned inside the flat buffer
    # backward accumulates: a second call doubles the gradient
    g2 = torch.from_numpy(grad).cuda()
    h.train_backward(dev(d_eps), g2)
    assert np.abs(g2.cpu().numpy() - 2 * grad).max() <= 1e-3 * np.abs(grad).max()


@pytest.mark.parametrize('tile_rows', [32, 64])
def test_data_gradient_kernel_all_modes(tile_rows):
    """k_dgrad_split on both tile sizes: ragged M (partial last tile), one and two sources, accumulate, the division and
    SiLU' epilogues, against fp64; the single-piece mode against the fp64 product of the bf16-rounded operands."""
    h = make_handle(ModelConfig(hidden_nf=64, n_layers=1))
    g = torch.Generator().manual_seed(3)
    W = (torch.rand(2, 256, 256, generator=g) / 8 - 1 / 16).cuda()
    dsilu = lambda v: torch.sigmoid(v) * (1 + v * (1 - torch.sigmoid(v)))
    for M in (1, 31, 64, 257, 1000):
        A0, A1 = torch.randn(M, 256, generator=g).cuda(), torch.randn(M, 256, generator=g).cuda()
        pre, Y0 = torch.randn(M, 256, generator=g).cuda(), torch.randn(M, 256, generator=g).cuda()
        want1 = A0.double() @ W[0].double()
        got = h.debug_dgrad(A0, W[0], tile_rows=tile_rows)
        assert (got.double() - want1).abs().max() <= 2e-6 * float(want1.abs().max()) + 1e-6, M
        want2 = Y0.double() + (A0.double() @ W[0].double() + A1.double() @ W[1].double()) / 100.0 * dsilu(pre.double())
        Y = Y0.clone()
        h.debug_dgrad(A0, W[0], A1, W[1], Y=Y, accumulate=True, div=100.0, pre=pre, tile_rows=tile_rows)
        assert (Y.double() - want2).abs().max() <= 3e-6 * float(want2.abs().max()) + 1e-6, M
        bf = lambda t: t.to(torch.bfloat16).double()
        want3 = bf(A0) @ bf(W[0])
        got3 = h.debug_dgrad(A0, W[0], pieces=1, tile_rows=tile_rows)
        assert (got3.double() - want3).abs().max() <= 2e-5 * float(want3.abs().max()) + 1e-6, M


@pytest.mark.parametrize('mode', [0, 1, 3])
def test_weight_gradient_launch(mode):
    """k_wgrad_group (fp32 instruction), bf16 operands (mode 1: k_wgrad_split<1>, and k_wgrad_split128<1> from K = 65536 up) and three
    bf16 pieces per operand (mode 3: k_wgrad_split128<3> / k_wgrad_split<3>, fp32-accurate): dW += dY^T X with the bias gradient folded
    in, on 256 x 256, on a 64-tile-only shape, on a shape only the generic kernel takes, K not a multiple of anything."""
    h = make_handle(ModelConfig(hidden_nf=64, n_layers=1))
    g = torch.Generator().manual_seed(4)
    for K, M, N in [(3001, 256, 256), (130, 64, 128), (777, 40, 24), (70001, 128, 256)]:
        dY, X = torch.randn(K, M, generator=g).cuda(), torch.randn(K, N, generator=g).cuda()
        dW0, db0 = torch.randn(M, N, generator=g).cuda(), torch.randn(M, generator=g).cuda()
        dW, db = dW0.clone(), db0.clone()
        h.debug_wgrad(dY, X, dW, db, mode=mode)
        torch.cuda.synchronize()
        r = (lambda t: t.to(torch.bfloat16).double()) if mode == 1 else (lambda t: t.double())
        want = dW0.double() + r(dY).t() @ r(X)
        tol = 2e-5 if mode == 1 else 1e-5
        assert (dW.double() - want).abs().max() <= tol * float(want.abs().max()), (K, M, N)
        assert (db.double() - (db0.double() + dY.double().sum(0))).abs().max() <= 1e-5 * float(dY.abs().sum(0).max()), (K, M, N)


# ------------------------------------------------------------------ the reference's training step (G11)
def build_trainer(lr=1e-3, inv_sublayers=1, aggregation_method='sum'):
    from argparse import Namespace
    from cmdgen_amd.lightning_modules import PharPocketDDPM
    from cmdgen_amd.training import HipTrainer
    g6 = load_golden('g6_loss.npz')
    cfg, sd, phar, pocket, hist = loss_case(g6)
    if inv_sublayers != 1 or aggregation_method != 'sum':
        import dataclasses
        cfg = dataclasses.replace(cfg, inv_sublayers=inv_sublayers, aggregation_method=aggregation_method)
        sd = make_state_dict(cfg, seed=int(g6['meta'][4]), coord_gain=1.0)
    hp = dict(outdir='out', dataset='crossdock', datadir='data', batch_size=4, lr=lr,
              egnn_params=Namespace(device='cuda', edge_cutoff=6.0, joint_nf=32, hidden_nf=cfg.hidden_nf,
                                    n_layers=cfg.n_layers, attention=True, tanh=True, norm_constant=1, inv_sublayers=inv_sublayers,
                                    sin_embedding=False, aggregation_method=aggregation_method, normalization_factor=100),
              diffusion_params=Namespace(diffusion_steps=500, diffusion_noise_schedule='polynomial_2',
                                         diffusion_noise_precision=1e-5, diffusion_loss_type='l2',
                                         normalize_factors=[1, 4]),
              num_workers=0, augment_noise=0, augment_rotation=False, clip_grad=True, eval_epochs=50,
              eval_params=Namespace(n_eval_samples=100, eval_batch_size=100), mode='pocket_conditioning',
              node_histogram=hist, pocket_representation='CA')
    model = PharPocketDDPM(**hp)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model = model.cuda()
    data = {'phar_coords': phar['x'], 'phar_one_hot': phar['one_hot'], 'num_phar_atoms': phar['size'],
            'phar_mask': phar['mask'], 'pocket_c_alpha': pocket['x'], 'pocket_one_hot': pocket['one_hot'],
            'num_pocket_nodes': pocket['size'], 'pocket_mask': pocket['mask']}
    return model, HipTrainer(model), data, g6


def test_training_step_matches_reference_gradients_and_optimizer():
    g = load_golden('g11_train.npz')
    model, tr, data, g6 = build_trainer()
    t_int, eps = dev(g6['t_int']), [dev(g6['eps0'])]
    loss, nll, info = tr.loss_and_grad(data, t_int=t_int, eps=eps)
    assert abs(float(loss) - float(g['step0/loss'])) < 2e-6
    assert np.abs(nll.cpu().numpy() - g['step0/nll']).max() < 1e-5
    grad = tr.grad.cpu().numpy()
    n = 0
    for key, want in g.items():
        if key.startswith('grad/') and key != 'grad/gamma.gamma':
            off, cnt = tr.h.param_offset(key[len('grad/dynamics.'):])
            got = grad[off:off + cnt].reshape(want.shape)
            assert np.abs(got - want).max() <= GRAD_TOL * max(float(np.abs(want).max()), 1e-6), key
            n += cnt
    assert n <= grad.size
    # three optimizer steps as the golden script took them: free, forced clip at half the norm, queue-driven
    for step in range(3):
        if step > 0:
            loss, nll, info = tr.loss_and_grad(data, t_int=t_int, eps=eps)
            assert abs(float(loss) - float(g[f'step{step}/loss'])) < 2e-5
        forced = float(g['step1/max_grad_norm']) if step == 1 else None
        grad_norm, mx = tr.optimizer_step(forced)
        assert abs(grad_norm - float(g[f'step{step}/grad_norm'])) <= 1e-4 * float(g[f'step{step}/grad_norm'])
        assert abs(mx - float(g[f'step{step}/max_grad_norm'])) <= 1e-4 * float(g[f'step{step}/max_grad_norm'])
    sd_after = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    worst = 0.0
    for key, want in g.items():
        if key.startswith('param_after3/') and 'gamma' not in key:
            name = key[len('param_after3/'):]
            got = sd_after['ddpm.' + name]
            g0 = g['grad/' + name]
            sig = np.abs(g0) > 1e-3 * np.abs(g0).max() if np.abs(g0).max() > 0 else np.zeros_like(g0, bool)
            # Adam normalises every element by its own gradient history: elements whose gradient is round-off
            # noise may legitimately move differently; the significant ones must agree closely
            if sig.any():
                worst = max(worst, float(np.abs(got - want)[sig].max()))
                assert np.abs(got - want)[sig].max() < 5e-5, name
            assert np.abs(got - want).max() <= 3 * 1e-3 * 2 + 1e-6, name
    assert worst > 0
    # the loss went down, and the sampler now runs on the updated weights (packed copy refreshed lazily)
    assert float(g['step2/loss']) < float(g['step0/loss'])
    pocket = {'x': data['pocket_c_alpha'].cuda(), 'one_hot': data['pocket_one_hot'].cuda(),
              'size': data['num_pocket_nodes'].cuda(), 'mask': data['pocket_mask'].cuda()}
    out = model.ddpm.sample_given_pocket(pocket, data['num_phar_atoms'], timesteps=5, seed=1)
    assert torch.isfinite(out[0]).all()


@pytest.mark.parametrize('loss_type', ['l2', 'vlb'])
def test_fused_loss_side_equals_the_tensor_op_path(loss_type):
    """cmdgen_train_noise / cmdgen_train_loss (three launches) against ConditionalDDPM.forward + PharPocketDDPM.forward
    evaluated with tensor operations on the same draws: every logged term, the per-sample nll, dL/d eps and the
    parameter gradient; one sample at t = 0 so that the L0 terms are exercised."""
    model, tr, data, g6 = build_trainer()
    model.loss_type = loss_type
    t_int = dev(g6['t_int']).clone()
    t_int[1] = 0
    eps = [dev(g6['eps0'])]
    tr.fused_loss = False
    loss_a, nll_a, info_a = tr.loss_and_grad(data, t_int=t_int, eps=eps)
    grad_a = tr.grad.clone()
    ctx = tr.ddpm._last_train_ctx
    tr.fused_loss = True
    assert tr._fused_ok()
    loss_b, nll_b, info_b = tr.loss_and_grad(data, t_int=t_int, eps=eps)
    f = tr._last_fused
    assert (f['net_out'] - ctx['net_out']).abs().max() <= 2e-5 * float(ctx['net_out'].abs().max())
    scale = max(1.0, float(nll_a.abs().max()))
    assert float((nll_a - nll_b).abs().max()) <= 2e-5 * scale, (nll_a, nll_b)
    assert abs(float(loss_a) - float(loss_b)) <= 2e-5 * scale
    assert set(info_a) == set(info_b)
    for k in info_a:
        a, b = float(info_a[k]), float(info_b[k])
        assert abs(a - b) <= 2e-5 * max(1.0, abs(a)), (k, a, b)
    assert float((tr.grad - grad_a).abs().max()) <= GRAD_TOL * float(grad_a.abs().max())
    assert float(f['terms'][1, 6]) > 0 and float(f['terms'][1, 8]) == 0          # the t = 0 sample carries L0, not L_t
    # without injected draws the fused path makes its own t and eps and still trains
    l0, _, _ = tr.loss_and_grad(data)
    assert np.isfinite(float(l0)) and torch.isfinite(tr.grad).all()


def _variant_trainer(mode, loss_type='l2'):
    """PharPocketDDPM + HipTrainer of another model variant on the fixtures' batches: 'joint' (G9) / 'pocket_conditioning_simple' (G6)."""
    from argparse import Namespace
    from helpers import joint_loss_case
    from cmdgen_amd.lightning_modules import PharPocketDDPM
    from cmdgen_amd.training import HipTrainer
    if mode == 'joint':
        g = load_golden('g9_joint.npz')
        cfg, sd, phar, pocket, hist = joint_loss_case(g)
    else:
        g = load_golden('g6_loss.npz')
        cfg, sd, phar, pocket, hist = loss_case(g)
    hp = dict(outdir='out', dataset='crossdock', datadir='data', batch_size=4, lr=1e-3,
              egnn_params=Namespace(device='cuda', edge_cutoff=6.0, joint_nf=32, hidden_nf=cfg.hidden_nf, n_layers=cfg.n_layers,
                                    attention=True, tanh=True, norm_constant=1, inv_sublayers=1, sin_embedding=False,
                                    aggregation_method='sum', normalization_factor=100),
              diffusion_params=Namespace(diffusion_steps=500, diffusion_noise_schedule='polynomial_2',
                                         diffusion_noise_precision=1e-5, diffusion_loss_type=loss_type, normalize_factors=[1, 4]),
              num_workers=0, augment_noise=0, augment_rotation=False, clip_grad=True, eval_epochs=50,
              eval_params=Namespace(n_eval_samples=100, eval_batch_size=100), mode=mode, node_histogram=hist, pocket_representation='CA')
    model = PharPocketDDPM(**hp)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model = model.cuda()
    data = {'phar_coords': phar['x'], 'phar_one_hot': phar['one_hot'], 'num_phar_atoms': phar['size'], 'phar_mask': phar['mask'],
            'pocket_c_alpha': pocket['x'], 'pocket_one_hot': pocket['one_hot'], 'num_pocket_nodes': pocket['size'],
            'pocket_mask': pocket['mask']}
    return model, HipTrainer(model), data, g


@pytest.mark.parametrize('mode,loss_type', [('joint', 'l2'), ('joint', 'vlb'), ('pocket_conditioning_simple', 'l2'),
                                            ('pocket_conditioning_simple', 'vlb')])
def test_fused_loss_side_of_the_joint_and_simple_variants(mode, loss_type):
    """cmdgen_train_noise_joint / cmdgen_train_loss_joint (mode 'joint': EnVariationalDiffusion.forward, en_diffusion.py:332-465) and
    cmdgen_train_noise / _loss on a no_com_projection handle (SimpleConditionalDDPM, conditional_model.py:481-525) against the same
    model's forward evaluated with tensor operations on the same draws: per-sample nll, every logged term, the network's outputs and the
    parameter gradient; one sample at t = 0."""
    model, tr, data, g = _variant_trainer(mode, loss_type)
    assert tr._variant() == ('joint' if mode == 'joint' else 'simple')
    if mode == 'joint':
        Nl, Np = len(data['phar_mask']), len(data['pocket_mask'])
        row = g['loss/train/noise'][0]
        eps = [(dev(row[:Nl * 11].reshape(Nl, 11).copy()), dev(row[Nl * 11:].reshape(Np, 23).copy()))]
        t_int = dev(g['loss/t_int']).clone()
    else:
        eps = [dev(g['eps0'])]
        t_int = dev(g['t_int']).clone()
    t_int[1] = 0
    tr.fused_loss = False
    loss_a, nll_a, info_a = tr.loss_and_grad(data, t_int=t_int, eps=eps)
    grad_a = tr.grad.clone()
    ctx = dict(tr.ddpm._last_train_ctx)
    tr.fused_loss = True
    assert tr._fused_ok()
    loss_b, nll_b, info_b = tr.loss_and_grad(data, t_int=t_int, eps=eps)
    f = tr._last_fused
    assert (f['net_out'] - ctx['net_out']).abs().max() <= 2e-5 * float(ctx['net_out'].abs().max())
    if mode == 'joint':
        assert (f['net_out_pocket'] - ctx['net_out_pocket']).abs().max() <= 2e-5 * float(ctx['net_out_pocket'].abs().max())
        assert (f['eps_t_pocket'] - ctx['eps_t_pocket']).abs().max() <= 1e-6
    scale = max(1.0, float(nll_a.abs().max()))
    assert float((nll_a - nll_b).abs().max()) <= 2e-5 * scale, (nll_a, nll_b)
    assert abs(float(loss_a) - float(loss_b)) <= 2e-5 * scale
    assert set(info_a) == set(info_b), set(info_a) ^ set(info_b)
    for k in info_a:
        a, b = float(info_a[k]), float(info_b[k])
        assert abs(a - b) <= 2e-5 * max(1.0, abs(a)), (k, a, b)
    assert float((tr.grad - grad_a).abs().max()) <= GRAD_TOL * float(grad_a.abs().max())
    assert float(f['terms'][1, 6]) > 0 and float(f['terms'][1, 8]) == 0          # the t = 0 sample carries L0, not L_t
    l0, _, _ = tr.loss_and_grad(data)                                             # own t and draws
    assert np.isfinite(float(l0)) and torch.isfinite(tr.grad).all()


def test_pipelined_steps_equal_waiting_steps():
    """HipTrainer.pipelined (no wait for a step's own gradient norm; layout and per-sample table uploaded in stream order
    from pinned staging, the index arrays into the second of two device blocks; node counts from the batch's host copies)
    takes the same optimizer trajectory as the waiting mode on batches whose layout changes every step: same clipping
    bounds, same norms (collected one step late), same parameters."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location('bench_train', os.path.join(os.path.dirname(__file__), '..', 'tools', 'bench_train.py'))
    bt = importlib.util.module_from_spec(spec); spec.loader.exec_module(bt)
    batches = [bt.synthetic_batch(6, 900 + 10 * i, torch.device('cuda', 0)) for i in range(3)]
    gen = torch.Generator().manual_seed(5)
    draws = [(torch.randint(0, 501, (6, 1), generator=gen).float(), torch.randn((int(b['num_phar_atoms'].sum()), 11), generator=gen)) for b in batches]
    runs = []
    for pipelined in (False, True):
        model, tr, _, _ = build_trainer()
        tr.pipelined = pipelined
        for step in range(5):
            t_int, eps = draws[step % 3]
            info = tr.training_step(batches[step % 3], t_int=t_int, eps=[eps.cuda()])
            assert (info['grad_norm'] is None) == pipelined
        tr._collect_norm()
        torch.cuda.synchronize()
        assert tr.last_grad_norm is not None and tr.last_grad_norm > 0
        runs.append((tr.theta.clone(), list(tr.gradnorm_queue.items), tr.last_grad_norm))
    (th_a, q_a, n_a), (th_b, q_b, n_b) = runs
    assert len(q_a) == len(q_b) and np.allclose(q_a, q_b, rtol=1e-4), (q_a, q_b)
    assert abs(n_a - n_b) <= 1e-4 * n_a
    # float atomics order the gradient sums differently from run to run: Adam moves an element whose gradient is round-off
    # noise by up to lr per step either way, every other element agrees closely
    d = (th_a - th_b).abs()
    assert float(d.max()) <= 5 * 1e-3 * 2 and float(d.mean()) <= 2e-6, (float(d.max()), float(d.mean()))


def test_training_step_on_the_fp32_instruction_matches_the_split_engine():
    """cmdgen_set_gemm_mode(0) puts the whole step back on v_mfma_f32_32x32x2_f32 (forward tiles, k_sgemm data gradients, the
    unfused tail pass): same loss, same gradient as the default split-bf16 engine, and G11 holds for it too."""
    g = load_golden('g11_train.npz')
    model, tr, data, g6 = build_trainer()
    t_int, eps = dev(g6['t_int']), [dev(g6['eps0'])]
    loss_s, nll_s, _ = tr.loss_and_grad(data, t_int=t_int, eps=eps)
    grad_s = tr.grad.clone()
    assert tr.h.query('gemm_split') == 1
    tr.h.set_gemm_mode(False)
    assert tr.h.query('gemm_split') == 0
    loss_f, nll_f, _ = tr.loss_and_grad(data, t_int=t_int, eps=eps)
    grad_f = tr.grad.clone()
    tr.h.set_gemm_mode(True)
    assert abs(float(loss_s) - float(loss_f)) < 2e-6 and abs(float(loss_f) - float(g['step0/loss'])) < 2e-6
    assert float((grad_s - grad_f).abs().max()) <= GRAD_TOL * float(grad_f.abs().max())
    gf = grad_f.cpu().numpy()
    for key, want in g.items():
        if key.startswith('grad/') and key != 'grad/gamma.gamma':
            off, cnt = tr.h.param_offset(key[len('grad/dynamics.'):])
            assert np.abs(gf[off:off + cnt].reshape(want.shape) - want).max() <= GRAD_TOL * max(float(np.abs(want).max()), 1e-6), key


def test_training_gradient_at_bench_size_both_engines():
    """The benchmark's own shape (64 CrossDocked-shaped complexes: 3.7k nodes, 36k / 16k edges; H=256, L=5): the gradient of
    the split-engine step (fused data-gradient + tail kernels, forward edge kernels on the bf16 pipe) against the same step
    on the fp32 instruction with the stand-alone tail pass - every tensor to GRAD_TOL of its own scale - and the loss."""
    import importlib.util, os
    from argparse import Namespace
    from cmdgen_amd.lightning_modules import PharPocketDDPM
    from cmdgen_amd.training import HipTrainer
    spec = importlib.util.spec_from_file_location('bench_train', os.path.join(os.path.dirname(__file__), '..', 'tools', 'bench_train.py'))
    bt = importlib.util.module_from_spec(spec); spec.loader.exec_module(bt)
    cfg = ModelConfig()
    hp = dict(outdir='out', dataset='crossdock', datadir='data', batch_size=64, lr=1e-3,
              egnn_params=Namespace(device='cuda', edge_cutoff=6.0, joint_nf=32, hidden_nf=256, n_layers=5, attention=True, tanh=True,
                                    norm_constant=1, inv_sublayers=1, sin_embedding=False, aggregation_method='sum', normalization_factor=100),
              diffusion_params=Namespace(diffusion_steps=500, diffusion_noise_schedule='polynomial_2', diffusion_noise_precision=1e-5,
                                         diffusion_loss_type='l2', normalize_factors=[1, 4]),
              num_workers=0, augment_noise=0, augment_rotation=False, clip_grad=True, eval_epochs=50,
              eval_params=Namespace(n_eval_samples=100, eval_batch_size=100), mode='pocket_conditioning',
              node_histogram=np.ones((30, 500)), pocket_representation='CA')
    model = PharPocketDDPM(**hp)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(cfg, seed=0).items()}, strict=True)
    tr = HipTrainer(model.cuda())
    batch = bt.synthetic_batch(64, 7000, torch.device('cuda', 0))
    gen = torch.Generator().manual_seed(11)
    t_int = torch.randint(0, 501, (64, 1), generator=gen).float()
    eps = [torch.randn((int(batch['num_phar_atoms'].sum()), 11), generator=gen).cuda()]
    loss_s, _, _ = tr.loss_and_grad(batch, t_int=t_int, eps=eps)
    grad_s = tr.grad.clone()
    assert tr.h.query('train_edges') > 24576          # the big-list code paths
    tr.h.set_gemm_mode(False)
    loss_f, _, _ = tr.loss_and_grad(batch, t_int=t_int, eps=eps)
    grad_f = tr.grad.clone()
    tr.h.set_gemm_mode(True)
    assert abs(float(loss_s) - float(loss_f)) <= 2e-6 * max(1.0, abs(float(loss_f)))
    for name, p in tr.dyn.named_parameters():
        off, cnt = tr.h.param_offset(name)
        a, b = grad_s[off:off + cnt], grad_f[off:off + cnt]
        assert float((a - b).abs().max()) <= GRAD_TOL * max(float(b.abs().max()), 1e-6), name


@pytest.mark.parametrize('S,agg', [(1, 'sum'), (2, 'mean')])
def test_staged_backward_equals_single_pass(S, agg):
    """cmdgen_train_backward_stages (what the overlapped all-reduce drives) over any split of the stages 0..L+1 leaves
    the same flat gradient as the single call, and the chunks HipTrainer reduces are final when their stage is done
    (a stage is a BLOCK: with inv_sublayers > 1 it holds several GCLs)."""
    model, tr, data, g6 = build_trainer(inv_sublayers=S, aggregation_method=agg)
    t_int, eps = torch.from_numpy(g6['t_int']).cuda(), [torch.from_numpy(g6['eps0']).cuda()]
    tr.loss_and_grad(data, t_int=t_int, eps=eps)
    whole = tr.grad.clone()
    L = int(tr.dyn._cfg['n_layers'])
    # rebuild d_eps exactly as loss_and_grad does, via a second pass that records it
    seen = {}
    orig = tr.h.train_backward

    def rec(d_eps, grad, d_eps_q=None):
        seen['d_eps'] = d_eps.clone()
        return orig(d_eps, grad, d_eps_q)
    tr.h.train_backward = rec
    tr.loss_and_grad(data, t_int=t_int, eps=eps)
    tr.h.train_backward = orig
    assert torch.allclose(tr.grad, whole, rtol=0, atol=2e-6 * float(whole.abs().max()))    # float atomics: equal up to summation order
    d_eps = seen['d_eps']
    for split in ([(0, L + 1)], [(0, 0), (1, L), (L + 1, L + 1)], [(s, s) for s in range(L + 2)]):
        tr.grad.zero_()
        final_from = tr.theta.numel()
        for first, last in split:
            tr.h.train_backward_stages(d_eps, tr.grad, first, last)
            torch.cuda.synchronize()
            if 1 <= last <= L:                                           # blocks >= L-last are final now
                lo = tr.h.param_offset(f'egnn.e_block_{L - last}.gcl_0.edge_mlp.0.weight')[0]
                assert torch.allclose(tr.grad[lo:], whole[lo:], rtol=0, atol=1e-6 * float(whole.abs().max()))
                final_from = lo
        assert torch.allclose(tr.grad, whole, rtol=0, atol=2e-6 * float(whole.abs().max())), split
    chunks = tr.grad_chunks()
    assert chunks[0][2] == tr.theta.numel() and chunks[-1][1] == 0 and chunks[-1][0] == L + 1
    with pytest.raises(hip_backend.CmdgenError):
        tr.h.train_backward_stages(d_eps, tr.grad, 3, 2)


def test_sample_and_analyze_given_pocket_reports_type_kl(tmp_path):
    """Validation sampling (lightning_modules.py:337-382): n_samples pockets of a dataset, sampled phar types and
    returned pocket types against the training histograms.  The pocket types come back exactly as they went in, so
    their KL equals that of the dataset's own composition."""
    from cmdgen_amd.dataset import ProcessedLigandPharPocketDataset, write_synthetic_npz
    model, tr, data, g6 = build_trainer()
    write_synthetic_npz(tmp_path / 'val.npz', n_complexes=6, seed=3)
    ds = ProcessedLigandPharPocketDataset(tmp_path / 'val.npz')
    model.eval()
    torch.manual_seed(0)
    out = model.sample_and_analyze_given_pocket(5, ds, batch_size=2, timesteps=5)
    assert set(out) == {'kl_div_atom_types', 'kl_div_residue_types'} and all(np.isfinite(v) for v in out.values())
    types = np.concatenate([ds[i % len(ds)]['pocket_one_hot'].argmax(1).numpy() for i in (0, 1, 2, 3, 4)])
    assert abs(out['kl_div_residue_types'] - model._type_kl(model.dataset_info['aa_hist'], model.dataset_info['aa_encoder'], types)) < 1e-9


def test_training_reduces_the_loss_on_a_fixed_batch():
    """20 steps with fresh t / noise per step on one synthetic batch: the objective must fall (sanity of the whole loop)."""
    model, tr, data, g6 = build_trainer()
    tr.lr = 2e-3
    torch.manual_seed(0)
    losses = [float(tr.training_step(data)['loss']) for _ in range(30)]
    assert np.isfinite(losses).all()
    assert np.mean(losses[-5:]) < 0.8 * np.mean(losses[:5]), losses


# ------------------------------------------------------------------ bf16 GEMM operands (opt-in mixed precision)
@pytest.mark.parametrize('ta,tb', [(False, True), (False, False), (True, False)])
def test_bf16_gemm_equals_fp32_product_of_rounded_operands(ta, tb):
    """bf16 mode rounds the operands (nearest-even) and accumulates in fp32: it must match an fp64 product of the
    bf16-rounded operands to fp32 accuracy - this also pins the operand layout of v_mfma_f32_32x32x16_bf16."""
    h = make_handle(ModelConfig(hidden_nf=64, n_layers=1))
    g = torch.Generator().manual_seed(2)
    for M, N, K in [(64, 64, 64), (130, 70, 96), (257, 256, 1000), (33, 5, 77)]:
        A = torch.randn((K, M) if ta else (M, K), generator=g).cuda()
        B = torch.randn((N, K) if tb else (K, N), generator=g).cuda()
        Ar, Br = A.to(torch.bfloat16).double(), B.to(torch.bfloat16).double()
        want = (Ar.t() if ta else Ar) @ (Br.t() if tb else Br)
        got = h.debug_sgemm(A, B, ta=ta, tb=tb, bf16=True)
        assert (got.double() - want).abs().max() <= 2e-5 * max(1.0, float(want.abs().max())), (M, N, K)
        exact = (A.double().t() if ta else A.double()) @ (B.double().t() if tb else B.double())
        assert (got.double() - exact).abs().max() > 1e-4 * float(exact.abs().max())      # it really is the bf16 path


def test_bf16_training_gradients_close_and_loss_falls():
    g = load_golden('g11_train.npz')
    from cmdgen_amd.training import HipTrainer
    model, tr, data, g6 = build_trainer()
    tr.gemm_dtype = 'bf16'
    t_int, eps = dev(g6['t_int']), [dev(g6['eps0'])]
    loss, nll, info = tr.loss_and_grad(data, t_int=t_int, eps=eps)
    assert abs(float(loss) - float(g['step0/loss'])) < 5e-3 * abs(float(g['step0/loss']))
    grad = tr.grad.cpu().numpy()
    # direction and size of the whole gradient agree with the fp32 reference to bf16 accuracy
    ref = np.zeros_like(grad)
    for key, want in g.items():
        if key.startswith('grad/') and key != 'grad/gamma.gamma':
            off, cnt = tr.h.param_offset(key[len('grad/dynamics.'):])
            ref[off:off + cnt] = want.reshape(-1)
    cos = float(np.dot(grad, ref) / (np.linalg.norm(grad) * np.linalg.norm(ref)))
    assert cos > 0.999, cos
    assert abs(np.linalg.norm(grad) / np.linalg.norm(ref) - 1.0) < 2e-2
    assert np.abs(grad - ref).max() > 1e-6 * np.abs(ref).max()            # not the fp32 path
    tr.lr = 2e-3
    torch.manual_seed(0)
    losses = [float(tr.training_step(data)['loss']) for _ in range(30)]
    assert np.isfinite(losses).all() and np.mean(losses[-5:]) < 0.8 * np.mean(losses[:5]), losses


def test_train_driver_end_to_end(tmp_path):
    """python -m cmdgen_amd.train --config cfg.yml: the reference's config keys, NPZ datasets, epochs, validation loss,
    Lightning-format checkpoints the sampler loads, --resume continuing the optimizer state."""
    import json
    import yaml
    from cmdgen_amd import train as train_cli
    from cmdgen_amd.dataset import write_synthetic_npz
    from cmdgen_amd.lightning_modules import PharPocketDDPM
    data = tmp_path / 'data'; data.mkdir()
    write_synthetic_npz(data / 'train.npz', n_complexes=12, seed=1)
    write_synthetic_npz(data / 'val.npz', n_complexes=5, seed=2)
    np.save(data / 'size_distribution.npy', np.ones((30, 70)))
    cfg = {'run_name': 'unit', 'logdir': str(tmp_path / 'logs'), 'wandb_params': {'mode': 'disabled'}, 'dataset': 'crossdock',
           'datadir': str(data), 'enable_progress_bar': False, 'num_sanity_val_steps': 0, 'mode': 'pocket_conditioning',
           'pocket_representation': 'CA', 'batch_size': 4, 'lr': 1e-3, 'n_epochs': 2, 'num_workers': 0, 'gpus': 1,
           'clip_grad': True, 'augment_rotation': False, 'augment_noise': 0,
           'egnn_params': {'device': 'cuda', 'edge_cutoff': 6.0, 'joint_nf': 32, 'hidden_nf': 64, 'n_layers': 2,
                           'attention': True, 'tanh': True, 'norm_constant': 1, 'inv_sublayers': 1, 'sin_embedding': False,
                           'aggregation_method': 'sum', 'normalization_factor': 100},
           'diffusion_params': {'diffusion_steps': 500, 'diffusion_noise_schedule': 'polynomial_2',
                                'diffusion_noise_precision': 1e-5, 'diffusion_loss_type': 'l2', 'normalize_factors': [1, 4]},
           'eval_epochs': 2, 'eval_params': {'n_eval_samples': 7, 'eval_batch_size': 4}}
    cfg_path = tmp_path / 'cfg.yml'
    cfg_path.write_text(yaml.safe_dump(cfg))
    out = train_cli.main(['--config', str(cfg_path)])
    assert out['epochs'] == 2 and out['steps'] == 6 and np.isfinite(out['best_val'])
    ckdir = tmp_path / 'logs' / 'unit' / 'checkpoints'
    assert (ckdir / 'last.ckpt').exists() and len(list(ckdir.glob('best-model-epoch=*.ckpt'))) == 1
    rows = [json.loads(x) for x in (tmp_path / 'logs' / 'unit' / 'metrics.jsonl').read_text().splitlines()]
    assert [r['epoch'] for r in rows] == [0, 1] and all(np.isfinite(r['loss/val']) for r in rows)
    # validation sampling on rank 0 every eval_epochs (validation_epoch_end, lightning_modules.py:289-304): epoch 1 only
    assert 'kl_div_atom_types/val' not in rows[0]
    assert np.isfinite(rows[1]['kl_div_atom_types/val']) and np.isfinite(rows[1]['kl_div_residue_types/val'])
    assert rows[1]['kl_div_residue_types/val'] >= 0 and rows[1]['evaluation_s/val'] > 0
    # the sampler loads what the trainer wrote (Lightning checkpoint format) and the weights did move
    best = next(ckdir.glob('best-model-epoch=*.ckpt'))
    model = PharPocketDDPM.load_from_checkpoint(str(best), map_location='cuda').cuda()
    fresh = PharPocketDDPM(**model.hparams)
    assert any(not torch.equal(a.cpu(), b.cpu()) for a, b in zip(model.state_dict().values(), fresh.state_dict().values()))
    # resume: one more epoch, optimizer step counter continues
    cfg['n_epochs'] = 3
    cfg_path.write_text(yaml.safe_dump(cfg))
    out2 = train_cli.main(['--config', str(cfg_path), '--resume', str(ckdir / 'last.ckpt')])
    assert out2['epochs'] == 1 and out2['steps'] == 9


# ------------------------------------------------------------------ joint model
def test_joint_training_gradients_vs_oracle_autograd():
    """mode 'joint': pocket nodes move and are denoised too (residue decoder, all-node velocity with its centre of mass
    removed); gradient of the l2 training loss (en_diffusion.py:332-465 + lightning_modules.py:198-217) vs autograd
    through the oracle, t = 0 and t = T in the batch, draws pinned."""
    from argparse import Namespace
    from helpers import joint_loss_case, JointNoiseTape
    from cmdgen_amd.lightning_modules import PharPocketDDPM
    from cmdgen_amd.training import HipTrainer
    G9 = load_golden('g9_joint.npz')
    cfg, sd, phar, pocket, hist = joint_loss_case(G9)
    hp = dict(outdir='out', dataset='crossdock', datadir='data', batch_size=4, lr=1e-3,
              egnn_params=Namespace(device='cuda', edge_cutoff=6.0, joint_nf=32, hidden_nf=cfg.hidden_nf, n_layers=cfg.n_layers,
                                    attention=True, tanh=True, norm_constant=1, inv_sublayers=1, sin_embedding=False,
                                    aggregation_method='sum', normalization_factor=100),
              diffusion_params=Namespace(diffusion_steps=500, diffusion_noise_schedule='polynomial_2',
                                         diffusion_noise_precision=1e-5, diffusion_loss_type='l2', normalize_factors=[1, 4]),
              num_workers=0, augment_noise=0, augment_rotation=False, clip_grad=True, eval_epochs=50,
              eval_params=Namespace(n_eval_samples=100, eval_batch_size=100), mode='joint', node_histogram=hist,
              pocket_representation='CA')
    model = PharPocketDDPM(**hp)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model = model.cuda()
    tr = HipTrainer(model)
    data = {'phar_coords': phar['x'], 'phar_one_hot': phar['one_hot'], 'num_phar_atoms': phar['size'], 'phar_mask': phar['mask'],
            'pocket_c_alpha': pocket['x'], 'pocket_one_hot': pocket['one_hot'], 'num_pocket_nodes': pocket['size'],
            'pocket_mask': pocket['mask']}
    Nl, Np = len(phar['mask']), len(pocket['mask'])
    row = G9['loss/train/noise'][0]
    eps = [(dev(row[:Nl * 11].reshape(Nl, 11)), dev(row[Nl * 11:].reshape(Np, 23)))]
    loss, nll, info = tr.loss_and_grad(data, t_int=dev(G9['loss/t_int']), eps=eps)
    # oracle: same loss with autograd
    p = ref_cpu.to_torch_params(sd)
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items() if k.startswith('dynamics.')}
    p2 = dict(p); p2.update(leaves)
    tape = JointNoiseTape(G9['loss/train/noise'], Nl, Np)
    terms = ref_cpu.joint_ddpm_forward(p2, cfg.as_dict(), phar, pocket, torch.from_numpy(G9['loss/t_int']), tape, training=True,
                                       histogram=hist)
    want_nll = ref_cpu.nll_from_terms(terms, cfg.as_dict(), phar['size'], pocket['size'], training=True)
    want_nll.mean(0).backward()
    assert np.abs(nll.cpu().numpy() - want_nll.detach().numpy()).max() < 2e-5 * max(1.0, float(want_nll.detach().abs().max()))
    grad = tr.grad.cpu().numpy()
    n_nonzero = 0
    for name, leaf in leaves.items():
        off, cnt = tr.h.param_offset(name[len('dynamics.'):])
        g_want = np.zeros(cnt, np.float32) if leaf.grad is None else leaf.grad.numpy().reshape(-1)
        scale = max(float(np.abs(g_want).max()), 1e-6)
        assert np.abs(grad[off:off + cnt] - g_want).max() <= GRAD_TOL * scale, (name, scale)
        n_nonzero += int(np.abs(g_want).max() > 0)
    assert n_nonzero == len(leaves)                 # the residue decoder is trained in joint mode
    torch.manual_seed(0)
    tr.lr = 2e-3
    losses = [float(tr.training_step(data)['loss']) for _ in range(40)]
    assert np.isfinite(losses).all() and np.mean(losses[-8:]) < 0.93 * np.mean(losses[:8]), losses    # fresh t / noise each step: noisy


def test_one_handle_many_layouts_soak():
    """Workspaces are capacity-based: a handle that has seen a big batch reuses its buffers for every later layout that
    fits.  Walk one handle through shrinking / growing ragged layouts, mixing evaluations, graph and eager chains and
    training forward/backward, and check every result against the oracle (stale state from a previous layout - graphs,
    index arrays, the agg-is-zero invariant, activation stores - would show up here)."""
    cfg = ModelConfig(hidden_nf=64, n_layers=2)
    sd = make_state_dict(cfg, seed=77, coord_gain=1.0)
    p = ref_cpu.to_torch_params(sd)
    h = make_handle(cfg)
    h.load_state_dict(sd)
    theta = flat_theta(h, sd)
    rng = np.random.Generator(np.random.PCG64(5))
    for it, B in enumerate([6, 2, 5, 1, 7, 3, 6, 2]):
        pb, pm, xh_phar, xh_pocket, t = case_inputs(cfg, B, 910000 + 37 * it, rng)
        h.set_layout(pb.num_nodes_phar, pb.size)
        with torch.no_grad():
            want, _ = ref_cpu.dynamics_forward(p, cfg.as_dict(), torch.from_numpy(xh_phar), torch.from_numpy(xh_pocket),
                                               torch.from_numpy(t), torch.from_numpy(pm), torch.from_numpy(pb.mask))
        want = want.numpy()
        tol = 2e-5 * max(1.0, float(np.abs(want).max()))
        got, _ = h.dynamics_forward(dev(xh_phar), dev(xh_pocket), dev(t))
        assert np.abs(got.cpu().numpy() - want).max() <= tol, (it, B)
        if it % 2 == 0:         # training path on the same handle and layout
            eps = h.train_forward(theta, dev(xh_phar), dev(xh_pocket), dev(t))
            assert np.abs(eps.cpu().numpy() - want).max() <= tol, (it, B)
            g1 = torch.zeros_like(theta); g2 = torch.zeros_like(theta)
            d_eps = dev(rng.normal(size=want.shape).astype(np.float32))
            h.train_backward(d_eps, g1)
            h.train_forward(theta, dev(xh_phar), dev(xh_pocket), dev(t))
            h.train_backward(d_eps, g2)
            assert float((g1 - g2).abs().max()) <= 1e-3 * float(g1.abs().max())      # repeatable up to atomics order
        # a short chain, graph replay vs eager launches with the same injected noise
        K = 4
        noise = dev(rng.normal(size=(K + 2, len(pm), 11)).astype(np.float32))
        px, poh = dev(pb.x), dev(pb.one_hot)
        a = h.sample_chain(px, poh, K, noise=noise, use_graph=True)
        b = h.sample_chain(px, poh, K, noise=noise, use_graph=False)
        assert torch.equal(a[0][:, 3:], b[0][:, 3:])
        sc = max(1.0, float(a[0][:, :3].abs().max()))
        assert float((a[0][:, :3] - b[0][:, :3]).abs().max()) <= 1e-4 * sc, (it, B)
        st = h.chain_status()
        assert st['max_rel_com_error'] < 1e-2


def test_simple_conditional_training_gradients_vs_oracle_autograd():
    """mode 'pocket_conditioning_simple' (SimpleConditionalDDPM: pocket-centred frame, no COM projection) trains through
    the same path; gradient of the l2 training loss vs autograd through the oracle."""
    from argparse import Namespace
    from cmdgen_amd.lightning_modules import PharPocketDDPM
    from cmdgen_amd.training import HipTrainer
    g6 = load_golden('g6_loss.npz')
    cfg, sd, phar, pocket, hist = loss_case(g6)
    cfg.no_com_projection = True
    hp = dict(outdir='out', dataset='crossdock', datadir='data', batch_size=4, lr=1e-3,
              egnn_params=Namespace(device='cuda', edge_cutoff=6.0, joint_nf=32, hidden_nf=cfg.hidden_nf, n_layers=cfg.n_layers,
                                    attention=True, tanh=True, norm_constant=1, inv_sublayers=1, sin_embedding=False,
                                    aggregation_method='sum', normalization_factor=100),
              diffusion_params=Namespace(diffusion_steps=500, diffusion_noise_schedule='polynomial_2',
                                         diffusion_noise_precision=1e-5, diffusion_loss_type='l2', normalize_factors=[1, 4]),
              num_workers=0, augment_noise=0, augment_rotation=False, clip_grad=True, eval_epochs=50,
              eval_params=Namespace(n_eval_samples=100, eval_batch_size=100), mode='pocket_conditioning_simple',
              node_histogram=hist, pocket_representation='CA')
    model = PharPocketDDPM(**hp)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model = model.cuda()
    tr = HipTrainer(model)
    data = {'phar_coords': phar['x'], 'phar_one_hot': phar['one_hot'], 'num_phar_atoms': phar['size'], 'phar_mask': phar['mask'],
            'pocket_c_alpha': pocket['x'], 'pocket_one_hot': pocket['one_hot'], 'num_pocket_nodes': pocket['size'],
            'pocket_mask': pocket['mask']}
    t_int = torch.tensor([[3.], [137.], [500.], [42.]])          # no t = 0 here: keeps every evaluation clear of the cutoff
    loss, nll, info = tr.loss_and_grad(data, t_int=t_int.cuda(), eps=[dev(g6['eps0'])])
    p = ref_cpu.to_torch_params(sd)
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items() if k.startswith('dynamics.')}
    p2 = dict(p); p2.update(leaves)
    terms = ref_cpu.ddpm_forward(p2, cfg.as_dict(), phar, pocket, t_int, [torch.from_numpy(g6['eps0'])], training=True,
                                 histogram=hist)
    want = ref_cpu.nll_from_terms(terms, cfg.as_dict(), phar['size'], pocket['size'], training=True)
    want.mean(0).backward()
    assert np.abs(nll.cpu().numpy() - want.detach().numpy()).max() < 2e-5 * max(1.0, float(want.detach().abs().max()))
    grad = tr.grad.cpu().numpy()
    for name, leaf in leaves.items():
        off, cnt = tr.h.param_offset(name[len('dynamics.'):])
        g_want = np.zeros(cnt, np.float32) if leaf.grad is None else leaf.grad.numpy().reshape(-1)
        assert np.abs(grad[off:off + cnt] - g_want).max() <= GRAD_TOL * max(float(np.abs(g_want).max()), 1e-6), name


def test_vlb_objective_gradients_vs_oracle_autograd():
    """diffusion_loss_type 'vlb' (predefined schedule): loss_t = -T/2 SNR_weight error_t, un-normalised L0, minus
    delta_log_px and log_pN (lightning_modules.py:209-231); gradient vs autograd through the oracle."""
    model, tr, data, g6 = build_trainer()
    model.loss_type = 'vlb'
    cfg, sd, phar, pocket, hist = loss_case(g6)
    t_int, eps = dev(g6['t_int']), [dev(g6['eps0'])]
    loss, nll, info = tr.loss_and_grad(data, t_int=t_int, eps=eps)
    p = ref_cpu.to_torch_params(sd)
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items() if k.startswith('dynamics.')}
    p2 = dict(p); p2.update(leaves)
    terms = ref_cpu.ddpm_forward(p2, cfg.as_dict(), phar, pocket, torch.from_numpy(g6['t_int']), [torch.from_numpy(g6['eps0'])],
                                 training=True, histogram=hist)
    want = ref_cpu.nll_from_terms(terms, cfg.as_dict(), phar['size'], pocket['size'], training=True, loss_type='vlb')
    want.mean(0).backward()
    assert np.abs(nll.cpu().numpy() - want.detach().numpy()).max() < 2e-5 * max(1.0, float(want.detach().abs().max()))
    grad = tr.grad.cpu().numpy()
    for name, leaf in leaves.items():
        off, cnt = tr.h.param_offset(name[len('dynamics.'):])
        g_want = np.zeros(cnt, np.float32) if leaf.grad is None else leaf.grad.numpy().reshape(-1)
        assert np.abs(grad[off:off + cnt] - g_want).max() <= GRAD_TOL * max(float(np.abs(g_want).max()), 1e-6), name


# ------------------------------------------------------------------ EGNN options in the training step (G19, round 5)
@pytest.mark.parametrize('case', ['s2_sum', 's1_mean', 's3_mean'])
def test_training_gradients_with_egnn_options_match_reference(case):
    """inv_sublayers > 1 (several GCLs per block, egnn_new.py:127-131, :152-154) and aggregation_method 'mean' (egnn_new.py:285-292)
    in the training step: loss, per-sample nll and the gradient of every tensor against the REAL reference's autograd
    (tests/golden/make_golden_r5.py; the G6 inputs); staged backward == single pass with several GCLs per stage."""
    g = load_golden('g19_train_options.npz')
    S, mean = [int(v) for v in g[f'{case}/options']]
    model, tr, data, g6 = build_trainer(inv_sublayers=S, aggregation_method='mean' if mean else 'sum')
    t_int, eps = dev(g6['t_int']), [dev(g6['eps0'])]
    loss, nll, info = tr.loss_and_grad(data, t_int=t_int, eps=eps)
    assert abs(float(loss) - float(g[f'{case}/loss'])) < 2e-6 * max(1.0, abs(float(g[f'{case}/loss'])))
    assert np.abs(nll.cpu().numpy() - g[f'{case}/nll']).max() < 1e-5
    grad = tr.grad.cpu().numpy()
    n = 0
    pre = f'{case}/grad/'
    for key, want in g.items():
        if key.startswith(pre) and key != pre + 'gamma.gamma':
            off, cnt = tr.h.param_offset(key[len(pre + 'dynamics.'):])
            got = grad[off:off + cnt].reshape(want.shape)
            assert np.abs(got - want).max() <= GRAD_TOL * max(float(np.abs(want).max()), 1e-6), key
            n += 1
    L = int(g6['meta'][1])
    assert n == 20 + 10 * S * L + 5 * L
    # one optimizer step runs and the sampler takes the updated weights
    tr.optimizer_step(None)
    pocket = {'x': data['pocket_c_alpha'].cuda(), 'one_hot': data['pocket_one_hot'].cuda(),
              'size': data['num_pocket_nodes'].cuda(), 'mask': data['pocket_mask'].cuda()}
    out = model.ddpm.sample_given_pocket(pocket, data['num_phar_atoms'], timesteps=5, seed=1)
    assert torch.isfinite(out[0]).all()


@pytest.mark.parametrize('S,agg', [(2, 'mean'), (2, 'sum'), (1, 'mean')])
def test_training_options_at_width_256_agree_across_engines(S, agg):
    """The same options on the width the fast kernels exist for (H = 256: half-engine forward with save hooks, fused tails, weight
    gradients beside the data gradients): the step's gradient against the same step on the fp32 instruction with the stand-alone
    passes (the path G19 pins against the reference), every tensor to GRAD_TOL of its own scale - and against the oracle's autograd loss."""
    import importlib.util, os
    from argparse import Namespace
    from cmdgen_amd.lightning_modules import PharPocketDDPM
    from cmdgen_amd.training import HipTrainer
    spec = importlib.util.spec_from_file_location('bench_train', os.path.join(os.path.dirname(__file__), '..', 'tools', 'bench_train.py'))
    bt = importlib.util.module_from_spec(spec); spec.loader.exec_module(bt)
    cfg = ModelConfig(n_layers=2, inv_sublayers=S, aggregation_method=agg)
    hp = dict(outdir='out', dataset='crossdock', datadir='data', batch_size=16, lr=1e-3,
              egnn_params=Namespace(device='cuda', edge_cutoff=6.0, joint_nf=32, hidden_nf=256, n_layers=2, attention=True, tanh=True,
                                    norm_constant=1, inv_sublayers=S, sin_embedding=False, aggregation_method=agg, normalization_factor=100),
              diffusion_params=Namespace(diffusion_steps=500, diffusion_noise_schedule='polynomial_2', diffusion_noise_precision=1e-5,
                                         diffusion_loss_type='l2', normalize_factors=[1, 4]),
              num_workers=0, augment_noise=0, augment_rotation=False, clip_grad=True, eval_epochs=50,
              eval_params=Namespace(n_eval_samples=100, eval_batch_size=100), mode='pocket_conditioning',
              node_histogram=np.ones((30, 500)), pocket_representation='CA')
    model = PharPocketDDPM(**hp)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(cfg, seed=0).items()}, strict=True)
    tr = HipTrainer(model.cuda())
    batch = bt.synthetic_batch(16, 7000, torch.device('cuda', 0))
    gen = torch.Generator().manual_seed(12)
    t_int = torch.randint(0, 501, (16, 1), generator=gen).float()
    eps = [torch.randn((int(batch['num_phar_atoms'].sum()), 11), generator=gen).cuda()]
    loss_s, _, _ = tr.loss_and_grad(batch, t_int=t_int, eps=eps)
    grad_s = tr.grad.clone()
    tr.h.set_gemm_mode(False)
    loss_f, _, _ = tr.loss_and_grad(batch, t_int=t_int, eps=eps)
    grad_f = tr.grad.clone()
    tr.h.set_gemm_mode(True)
    assert abs(float(loss_s) - float(loss_f)) <= 2e-6 * max(1.0, abs(float(loss_f)))
    n = 0
    for name, p in tr.dyn.named_parameters():
        off, cnt = tr.h.param_offset(name)
        a, b = grad_s[off:off + cnt], grad_f[off:off + cnt]
        assert float((a - b).abs().max()) <= GRAD_TOL * max(float(b.abs().max()), 1e-6), name
        n += 1
    assert n == 20 + 10 * S * 2 + 5 * 2


def test_staged_backward_with_the_second_stream_at_width_256():
    """H = 256: the weight gradients run on the handle's second stream and every staged call joins it - any split of the stages leaves the
    single call's gradient (the rotating buffers carry over between calls: GCL k of the pass uses the same ones whichever call runs it)."""
    import importlib.util, os
    from argparse import Namespace
    from cmdgen_amd.lightning_modules import PharPocketDDPM
    from cmdgen_amd.training import HipTrainer
    spec = importlib.util.spec_from_file_location('bench_train', os.path.join(os.path.dirname(__file__), '..', 'tools', 'bench_train.py'))
    bt = importlib.util.module_from_spec(spec); spec.loader.exec_module(bt)
    L = 3
    cfg = ModelConfig(n_layers=L)
    hp = dict(outdir='out', dataset='crossdock', datadir='data', batch_size=16, lr=1e-3,
              egnn_params=Namespace(device='cuda', edge_cutoff=6.0, joint_nf=32, hidden_nf=256, n_layers=L, attention=True, tanh=True,
                                    norm_constant=1, inv_sublayers=1, sin_embedding=False, aggregation_method='sum', normalization_factor=100),
              diffusion_params=Namespace(diffusion_steps=500, diffusion_noise_schedule='polynomial_2', diffusion_noise_precision=1e-5,
                                         diffusion_loss_type='l2', normalize_factors=[1, 4]),
              num_workers=0, augment_noise=0, augment_rotation=False, clip_grad=True, eval_epochs=50,
              eval_params=Namespace(n_eval_samples=100, eval_batch_size=100), mode='pocket_conditioning',
              node_histogram=np.ones((30, 500)), pocket_representation='CA')
    model = PharPocketDDPM(**hp)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(cfg, seed=0).items()}, strict=True)
    tr = HipTrainer(model.cuda())
    batch = bt.synthetic_batch(16, 7100, torch.device('cuda', 0))
    gen = torch.Generator().manual_seed(13)
    t_int = torch.randint(0, 501, (16, 1), generator=gen).float()
    eps = [torch.randn((int(batch['num_phar_atoms'].sum()), 11), generator=gen).cuda()]
    seen = {}
    orig = tr.h.train_backward

    def rec(d_eps, grad, d_eps_q=None):
        seen['d_eps'] = d_eps.clone()
        return orig(d_eps, grad, d_eps_q)
    tr.h.train_backward = rec
    tr.loss_and_grad(batch, t_int=t_int, eps=eps)
    tr.h.train_backward = orig
    whole = tr.grad.clone()
    assert tr.h.get_option('wgrad_stream') is None          # not set: the default, second stream on
    for split in ([(0, 0), (1, L), (L + 1, L + 1)], [(s, s) for s in range(L + 2)], [(0, 1), (2, L + 1)]):
        tr.grad.zero_()
        for first, last in split:
            tr.h.train_backward_stages(seen['d_eps'], tr.grad, first, last)
        torch.cuda.synchronize()
        assert torch.allclose(tr.grad, whole, rtol=0, atol=2e-6 * float(whole.abs().max())), split
    # and the serial pass (option wgrad_stream = 0) agrees with the two-stream one
    tr.h.set_option('wgrad_stream', 0)
    tr.grad.zero_()
    tr.h.train_backward(seen['d_eps'], tr.grad)
    torch.cuda.synchronize()
    assert torch.allclose(tr.grad, whole, rtol=0, atol=2e-6 * float(whole.abs().max()))


@pytest.mark.parametrize('opts', [{'wgrad_silu': 3}, {'wgrad_stream': 0, 'train_half': 0}, {'wgrad_split': 0}, {'train_node16': 0}, {'wgrad_k128': 1}])
def test_training_step_options_leave_the_gradient_where_it_is(opts):
    """Every launch choice of the round-5 training step (cmdgen_set_option) against the default on the width the fast kernels exist for:
    SiLU recomputed by the second-layer weight gradients, the serial pass without the half-engine forward, weight gradients on the fp32
    instruction, half-engine data gradients, one kernel for the small gradients, the generic node tiles, 128 x 128 weight-gradient tiles
    everywhere - the same loss and, tensor by tensor, the same gradient to GRAD_TOL of its scale."""
    import importlib.util, os
    from argparse import Namespace
    from cmdgen_amd.lightning_modules import PharPocketDDPM
    from cmdgen_amd.training import HipTrainer
    spec = importlib.util.spec_from_file_location('bench_train', os.path.join(os.path.dirname(__file__), '..', 'tools', 'bench_train.py'))
    bt = importlib.util.module_from_spec(spec); spec.loader.exec_module(bt)
    L = 2
    cfg = ModelConfig(n_layers=L)
    hp = dict(outdir='out', dataset='crossdock', datadir='data', batch_size=24, lr=1e-3,
              egnn_params=Namespace(device='cuda', edge_cutoff=6.0, joint_nf=32, hidden_nf=256, n_layers=L, attention=True, tanh=True,
                                    norm_constant=1, inv_sublayers=1, sin_embedding=False, aggregation_method='sum', normalization_factor=100),
              diffusion_params=Namespace(diffusion_steps=500, diffusion_noise_schedule='polynomial_2', diffusion_noise_precision=1e-5,
                                         diffusion_loss_type='l2', normalize_factors=[1, 4]),
              num_workers=0, augment_noise=0, augment_rotation=False, clip_grad=True, eval_epochs=50,
              eval_params=Namespace(n_eval_samples=100, eval_batch_size=100), mode='pocket_conditioning',
              node_histogram=np.ones((30, 500)), pocket_representation='CA')
    model = PharPocketDDPM(**hp)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(cfg, seed=0).items()}, strict=True)
    tr = HipTrainer(model.cuda())
    batch = bt.synthetic_batch(24, 7300, torch.device('cuda', 0))
    gen = torch.Generator().manual_seed(14)
    t_int = torch.randint(0, 501, (24, 1), generator=gen).float()
    eps = [torch.randn((int(batch['num_phar_atoms'].sum()), 11), generator=gen).cuda()]
    loss_a, _, _ = tr.loss_and_grad(batch, t_int=t_int, eps=eps)
    grad_a = tr.grad.clone()
    for k, v in opts.items():
        tr.h.set_option(k, v)
    loss_b, _, _ = tr.loss_and_grad(batch, t_int=t_int, eps=eps)
    grad_b = tr.grad.clone()
    assert abs(float(loss_a) - float(loss_b)) <= 2e-6 * max(1.0, abs(float(loss_a)))
    for name, p in tr.dyn.named_parameters():
        off, cnt = tr.h.param_offset(name)
        a, b = grad_a[off:off + cnt], grad_b[off:off + cnt]
        assert float((a - b).abs().max()) <= GRAD_TOL * max(float(a.abs().max()), 1e-6), (name, opts)


# ------------------------------------------------------------------ round 6: the big-list kernels against the ORACLE, the half engine's range, re-pack hygiene
def _bench_trainer(B, pipelined=False, sd_edit=None, lr=1e-3):
    """PharPocketDDPM (shipped hyper-parameters, seeded weights) + HipTrainer + one synthetic batch of B ragged C-alpha complexes (tools/bench_train.py)."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location('bench_train', os.path.join(os.path.dirname(__file__), '..', 'tools', 'bench_train.py'))
    bt = importlib.util.module_from_spec(spec); spec.loader.exec_module(bt)
    cfg, model, tr = bt.build_trainer(B, 'CA', 'fp32', torch.device('cuda', 0), pipelined=pipelined)
    sd = make_state_dict(cfg, seed=0)
    if sd_edit is not None:
        sd = sd_edit(dict(sd))
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        tr = bt.HipTrainer(model, gemm_dtype='fp32')
        tr.pipelined = pipelined
    tr.lr = lr
    return cfg, sd, model, tr, bt


def test_default_step_at_bench_size_matches_oracle_autograd():
    """The DEFAULT training step at the benchmark's size - 64 ragged complexes, H = 256, L = 5, ~36k message / ~16k coordinate edges: half-engine
    forward with save hooks, two / three streams, k_wgrad_split128, the fused k_dgrad_tail - against AUTOGRAD THROUGH THE ORACLE on the same batch,
    draws and time steps: per-sample nll and the gradient of every tensor at GRAD_TOL of its own scale.  (The other big-list tests compare
    engines with each other; this one pins the big-list kernels to the reference's arithmetic.)"""
    cfg, sd, model, tr, bt = _bench_trainer(64)
    from cmdgen_amd.synthetic import make_training_batch
    for first in range(7000, 9000, 100):  # a batch whose POCKETS keep every pair clear of the cutoff (the radius graph is a hard threshold; the pockets do not depend on the draw)
        nb = make_training_batch(64, first, 'CA')
        if min_cutoff_margin(nb['pocket_c_alpha'], nb['pocket_mask'], 6.0) > 1e-4:
            break
    batch = bt.synthetic_batch(64, first, torch.device('cuda', 0))
    nl_tot = int(batch['num_phar_atoms'].sum())
    hist = np.ones((30, 500))
    for seed in range(11, 60):            # ... and draws whose noised phar points do too
        gen = torch.Generator().manual_seed(seed)
        t_int = torch.randint(1, 501, (64, 1), generator=gen).float()
        eps0 = torch.randn((nl_tot, 11), generator=gen)
        loss, nll, info = tr.loss_and_grad(batch, t_int=t_int.cuda(), eps=[eps0.cuda()])
        z = tr._last_fused['z_t'][:, :3].cpu().numpy()
        q = tr._last_fused['xh_pocket'][:, :3].cpu().numpy()
        margin = min_cutoff_margin(np.concatenate([z, q]), np.concatenate([batch['phar_mask'].cpu().numpy(), batch['pocket_mask'].cpu().numpy()]), 6.0)
        if margin > 5e-5:
            break
    assert margin > 5e-5
    E, Ec = tr.h.query('train_edges'), tr.h.query('train_coord_edges')
    assert E > 24576 and tr.h.half_engine_active() and tr.h.get_option('train_half') is None and tr.h.get_option('wgrad_stream') is None
    grad = tr.grad.cpu().numpy()
    cpu = lambda k: batch[k].detach().cpu()
    phar = {'x': cpu('phar_coords'), 'one_hot': cpu('phar_one_hot'), 'size': cpu('num_phar_atoms'), 'mask': cpu('phar_mask')}
    pocket = {'x': cpu('pocket_c_alpha'), 'one_hot': cpu('pocket_one_hot'), 'size': cpu('num_pocket_nodes'), 'mask': cpu('pocket_mask')}
    p = ref_cpu.to_torch_params(sd)
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items() if k.startswith('dynamics.')}
    p2 = dict(p); p2.update(leaves)
    terms = ref_cpu.ddpm_forward(p2, cfg.as_dict(), phar, pocket, t_int, [eps0], training=True, histogram=hist)
    want = ref_cpu.nll_from_terms(terms, cfg.as_dict(), phar['size'], pocket['size'], training=True)
    want.mean(0).backward()
    w = want.detach().numpy()
    err_nll = float(np.abs(nll.cpu().numpy() - w).max())
    print(f'64 complexes, {E} / {Ec} edges, cutoff margin {margin:.1e}: per-sample nll max diff {err_nll:.2e} (|nll| max {np.abs(w).max():.3f})')
    assert err_nll <= 2e-5 * max(1.0, float(np.abs(w).max()))
    assert abs(float(loss) - float(w.mean())) <= 2e-6 * max(1.0, abs(float(w.mean())))
    worst = (0.0, None)
    for name, leaf in leaves.items():
        off, cnt = tr.h.param_offset(name[len('dynamics.'):])
        g_want = np.zeros(cnt, np.float32) if leaf.grad is None else leaf.grad.numpy().reshape(-1)
        rel = float(np.abs(grad[off:off + cnt] - g_want).max()) / max(float(np.abs(g_want).max()), 1e-6)
        if rel > worst[0]:
            worst = (rel, name)
        assert rel <= GRAD_TOL, (name, rel)
    print(f'worst tensor {worst[1]}: {worst[0]:.2e} of its own scale (tolerance {GRAD_TOL:.0e})')


def test_half_forward_sees_the_parameters_of_this_step():
    """With the half-engine forward only ONE fp32 fragment pack is refreshed per step (cmdgen_train.hip): after an optimizer update the forward must
    still multiply with the NEW weights in every kernel.  Two steps with a large learning rate on the default path against two steps with
    train_half = 0 (every pack refreshed): the second step's loss and gradient agree, and differ clearly from the first step's."""
    out = {}
    for th in (None, 0):
        cfg, sd, model, tr, bt = _bench_trainer(8, lr=2e-2)
        if th is not None:
            tr.h.set_option('train_half', th)
        tr.clip_grad = False
        batch = bt.synthetic_batch(8, 9100, torch.device('cuda', 0))
        gen = torch.Generator().manual_seed(3)
        t_int = torch.randint(1, 501, (8, 1), generator=gen).float().cuda()
        eps = [torch.randn((int(batch['num_phar_atoms'].sum()), 11), generator=gen).cuda()]
        l1, _, _ = tr.loss_and_grad(batch, t_int=t_int, eps=eps)
        g1 = tr.grad.clone()
        tr.optimizer_step()
        l2, _, _ = tr.loss_and_grad(batch, t_int=t_int, eps=eps)
        out[th] = (float(l1), g1.cpu().numpy(), float(l2), tr.grad.cpu().numpy().copy(), tr)
    (l1a, g1a, l2a, g2a, tra), (l1b, g1b, l2b, g2b, trb) = out[None], out[0]
    assert tra.h.half_engine_active()
    assert abs(l2a - l1a) > 1e-3 * abs(l1a), 'the update must move the loss for this test to mean anything'
    assert abs(l2a - l2b) <= 1e-4 * max(1.0, abs(l2b)), (l2a, l2b)
    for name, _p in tra.dyn.named_parameters():
        off, cnt = tra.h.param_offset(name)
        a, b = g2a[off:off + cnt], g2b[off:off + cnt]
        assert float(np.abs(a - b).max()) <= 5 * GRAD_TOL * max(float(np.abs(b).max()), 1e-6), name       # (two steps: the first step's rounding differences pass through an update with lr 2e-2)


def test_training_step_leaves_the_half_range_and_repeats_on_the_bf16_engine():
    """A model whose message MLP's hidden activation exceeds fp16's 65504 (fp32 keeps it finite): the half-engine forward yields a non-finite
    gradient, the device skips that update, the trainer switches the forward to the three-piece bf16 engine, repeats the batch and warns -
    parameters after the step equal those of a trainer that ran with train_half = 0 from the start."""
    def edit(sd):
        for s in ('.weight', '.bias'):
            k = 'ddpm.dynamics.egnn.e_block_1.gcl_0.edge_mlp.0' + s
            sd[k] = (sd[k] * 3.0e6).astype(np.float32)
        return sd
    res = {}
    for th in (None, 0):
        cfg, sd, model, tr, bt = _bench_trainer(8, sd_edit=edit)
        if th is not None:
            tr.h.set_option('train_half', th)
        batch = bt.synthetic_batch(8, 9100, torch.device('cuda', 0))
        gen = torch.Generator().manual_seed(3)
        t_int = torch.randint(1, 501, (8, 1), generator=gen).float().cuda()
        eps = [torch.randn((int(batch['num_phar_atoms'].sum()), 11), generator=gen).cuda()]
        theta0 = tr.theta.clone()
        if th is None:
            with pytest.warns(RuntimeWarning, match='half matrix engine'):
                info = tr.training_step(batch, t_int=t_int, eps=eps)
            assert tr.h.get_option('train_half') == 0 and tr.half_range_fallbacks == 1
        else:
            info = tr.training_step(batch, t_int=t_int, eps=eps)
        assert np.isfinite(float(info['loss'])) and np.isfinite(tr.last_grad_norm)
        assert tr.step_count == 1 and not torch.equal(tr.theta, theta0) and bool(torch.isfinite(tr.theta).all())
        res[th] = (float(info['loss']), tr.theta.cpu().numpy().copy(), float(tr.last_grad_norm))
    assert abs(res[None][0] - res[0][0]) <= 1e-6 * max(1.0, abs(res[0][0]))
    assert abs(res[None][2] - res[0][2]) <= 1e-4 * res[0][2]
    # (the first AdamW step moves every parameter by ~lr * g / |g|: where g is at the rounding level the two runs' updates may differ by up to 2 lr)
    d = np.abs(res[None][1] - res[0][1])
    assert d.max() <= 2.1e-3 and np.mean(d > 1e-6) < 1e-3, (float(d.max()), float(np.mean(d > 1e-6)))
