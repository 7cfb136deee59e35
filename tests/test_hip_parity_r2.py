"""GPU parity tests, round 2: the HIP path at BASELINE configs[4]'s real size, in a regime where the north-star's
"coords within 1e-4 RMS" holds as an ABSOLUTE bound, per-block intermediates, generate_phars against the
reference's own output, get_edges(mask, x) on arbitrary input, and per-call noise.

Tolerances (fp32): one evaluation max|d eps| <= 2e-5 * max(1, max|eps|); chains with injected noise: coordinate
RMS <= 1e-4 A ABSOLUTE in the bounded-|x| regime (G13), <= 1e-4 * max(1, max|x|) where random-init weights
inflate coordinates by 1/alpha_T ~ 316 (ulp(800 A) = 6e-5 A); one-hot types exact; radius graphs exact incl. order.
"""
import json
import os

import numpy as np
import pytest
import torch

from helpers import (GOLDEN, load_golden, cases_of, dynamics_case, rms, bounded_case, fullsize_chain_case, g5_case,
                     pocket_dict)
from cmdgen_amd import hip_backend
from cmdgen_amd.synthetic import ModelConfig, make_state_dict, make_pockets, min_cutoff_margin

pytestmark = pytest.mark.gpu
EVAL_TOL = 2e-5

G12 = load_golden('g12_fullsize.npz')
G13 = load_golden('g13_bounded.npz')
G5 = load_golden('g5_blocks.npz')
G7 = load_golden('g7_generate.npz')


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def new_handle(cfg, sd):
    h = hip_backend.Handle(cfg.as_dict(), 0)
    h.load_state_dict(sd)
    return h


def host_step_table(cfg, K):
    """Per-step scalars evaluated exactly as the reference does (bit-identical table, oracle-side helper)."""
    from oracle import ref_cpu
    table = ref_cpu.gamma_table(cfg.noise_schedule, cfg.timesteps, cfg.noise_precision)
    coef = ref_cpu.step_coefficients(table, cfg.timesteps, K).numpy()
    g0 = table[0]
    final = np.array([[float(torch.sqrt(torch.sigmoid(g0))), float(torch.sqrt(torch.sigmoid(-g0))),
                       float(torch.exp(0.5 * g0)), 0.0]], np.float32)
    return np.concatenate([coef, final])


# ----------------------------------------------------------------------------- configs[4] at its real size
@pytest.mark.parametrize('mt', [None, 16, 32, 64])
def test_fullsize_dynamics_forward_every_tile_size(mt, monkeypatch):
    """One evaluation at Np=366 / Nl=15 per sample (381-node samples: 145k candidate pairs each, ~13k edges),
    H=256, L=5, against the reference's output; tile sizes of all three MFMA kernels forced in turn."""
    name = 'dyn_fa366_b2'
    if mt is not None:
        for k in ('node_mt', 'edge_mt', 'coord_mt'):
            monkeypatch.setitem(hip_backend.DEFAULT_OPTIONS, k, mt)
    cfg, sd, inp = dynamics_case(G12, name)
    h = new_handle(cfg, sd)                                   # fresh handle: tile sizes are chosen in cmdgen_set_layout
    h.set_layout(G12[name + '/num_nodes_phar'], G12[name + '/pocket_size'])
    eps, _ = h.dynamics_forward(dev(inp['xh_phar']), dev(inp['xh_pocket']), dev(inp['t']))
    torch.cuda.synchronize()
    want = G12[name + '/eps_phar']
    assert np.array_equal(h.get_edges(), G12[name + '/edges'])              # 27k edges, exact incl. order
    err = float(np.abs(eps.cpu().numpy() - want).max())
    assert err <= EVAL_TOL * max(1.0, float(np.abs(want).max())), err
    h.close()


@pytest.mark.parametrize('use_graph', [False, True])
def test_fullsize_chain_matches_reference(use_graph):
    name = 'chain_fa366_K5'
    cfg, sd, pb, K = fullsize_chain_case(G12, name)
    h = new_handle(cfg, sd)
    h.set_layout(pb.num_nodes_phar, pb.size)
    h.set_step_table(K, host_step_table(cfg, K))
    xh_phar, xh_pocket, z_steps = h.sample_chain(dev(pb.x), dev(pb.one_hot), K, noise=dev(G12[name + '/noise']),
                                                 want_steps=True, use_graph=use_graph)
    st = h.chain_status()
    want = G12[name + '/xh_phar']
    for k in range(K):
        zs = G12[name + '/z_steps'][k]
        assert float(np.abs(z_steps[k].cpu().numpy() - zs).max()) <= 1e-4 * max(1.0, float(np.abs(zs).max())), k
    assert rms(xh_phar[:, :3].cpu().numpy(), want[:, :3]) <= 1e-4 * max(1.0, float(np.abs(want[:, :3]).max()))
    assert np.array_equal(xh_phar[:, 3:].cpu().numpy(), want[:, 3:])
    assert st['max_rel_com_error'] < 1e-2 and st['nan_resets'] == 0
    h.close()


def test_fullsize_batch256_properties():
    """BASELINE configs[4] itself: 256 full-atom pockets (366 atoms) + 15 phar points each, on-device Philox noise,
    a short strided chain: size-independent properties - zero phar COM, valid one-hot rows, pocket rigidly
    translated with its types untouched, counters consistent with the layout, graph replay == eager."""
    cfg = ModelConfig(residue_nf=11, timesteps=500)
    sd = make_state_dict(cfg, seed=0)
    h = new_handle(cfg, sd)
    B, K = 256, 4
    pb = make_pockets(B, 'full-atom', n_phar=15)
    h.set_layout(pb.num_nodes_phar, pb.size)
    px, poh = dev(pb.x), dev(pb.one_hot)
    h.reset_counters()
    xg, pg, _ = h.sample_chain(px, poh, K, noise=None, seed=77, pocket_ids=pb.pocket_index, use_graph=True)
    st = h.chain_status()
    c = h.counters()
    n = 366 + 15
    assert c['evaluations'] == K + 1 and c['nodes'] == (K + 1) * B * n
    assert (K + 1) * B * n <= c['edges'] <= (K + 1) * B * n * n and 0 <= c['edges_phar'] <= c['edges']
    assert c['edges'] / ((K + 1) * B) > 8000                          # ~13k edges per full-atom pocket
    assert st['max_rel_com_error'] < 1e-2 and st['nan_resets'] == 0
    xg, pg = xg.cpu().numpy(), pg.cpu().numpy()
    assert np.isfinite(xg).all() and np.isfinite(pg).all()
    oh = xg[:, 3:]
    assert np.all((oh == 0) | (oh == 1)) and np.all(oh.sum(1) == 1)
    com = np.add.reduceat(xg[:, :3], np.arange(0, B * 15, 15), axis=0)
    assert np.abs(com).max() < 5e-2
    shift = (pg[:, :3] - pb.x).reshape(B, 366, 3)
    assert np.abs(shift - shift[:, :1]).max() < 1e-3 * max(1.0, np.abs(shift).max())       # rigid translation per pocket
    assert np.array_equal(pg[:, 3:], pb.one_hot)
    xe, pe, _ = h.sample_chain(px, poh, K, noise=None, seed=77, pocket_ids=pb.pocket_index, use_graph=False)
    xe = xe.cpu().numpy()
    assert np.abs(xe[:, :3] - xg[:, :3]).max() <= 1e-4 * max(1.0, np.abs(xg[:, :3]).max())
    assert np.array_equal(xe[:, 3:], xg[:, 3:])
    # a shard of the batch, keyed by global pocket ids, reproduces its slice
    sub = make_pockets(8, 'full-atom', n_phar=15, first_index=96)
    h.set_layout(sub.num_nodes_phar, sub.size)
    xs, _, _ = h.sample_chain(dev(sub.x), dev(sub.one_hot), K, noise=None, seed=77, pocket_ids=sub.pocket_index)
    want = xg[96 * 15:104 * 15]
    assert np.abs(xs.cpu().numpy()[:, :3] - want[:, :3]).max() <= 1e-4 * max(1.0, np.abs(want[:, :3]).max())
    h.close()


# ----------------------------------------------------------------------------- absolute 1e-4 RMS
@pytest.mark.parametrize('use_graph', [False, True])
@pytest.mark.parametrize('name', cases_of(G13))
def test_bounded_regime_chain_absolute_rms(name, use_graph):
    """Chains of a model with noise_precision 0.05 (1/alpha_T = 4.5): |x| <= 17 A throughout, so the north-star's
    'coords within 1e-4 RMS of reference' is asserted as an ABSOLUTE bound in Angstrom - K=50 strided and the full
    K=T=500 chain, recorded reference noise."""
    cfg, sd, pb, K = bounded_case(G13, name)
    h = new_handle(cfg, sd)
    h.set_layout(pb.num_nodes_phar, pb.size)
    h.set_step_table(K, host_step_table(cfg, K))
    xh_phar, xh_pocket, _ = h.sample_chain(dev(pb.x), dev(pb.one_hot), K, noise=dev(G13[name + '/noise']), use_graph=use_graph)
    st = h.chain_status()
    want, wp = G13[name + '/xh_phar'], G13[name + '/xh_pocket']
    err = rms(xh_phar[:, :3].cpu().numpy(), want[:, :3])
    print(f'{name} graph={use_graph}: coordinate RMS vs reference {err:.3e} A at max|x| {np.abs(want[:, :3]).max():.1f} A')
    assert err <= 1e-4
    assert np.array_equal(xh_phar[:, 3:].cpu().numpy(), want[:, 3:])
    assert rms(xh_pocket[:, :3].cpu().numpy(), wp[:, :3]) <= 1e-4
    assert np.array_equal(xh_pocket[:, 3:].cpu().numpy(), wp[:, 3:])
    assert st['max_rel_com_error'] < 1e-2 and st['nan_resets'] == 0
    h.close()


def test_bounded_regime_full_T1000_chain_vs_oracle():
    """The headline shape's chain length (T = K = 1000) in the bounded regime, HIP vs oracle on the same draws,
    absolute 1e-4 A.  Seeds whose chain comes within 2e-5 A of the cutoff (hard-threshold graph) are skipped."""
    from oracle import ref_cpu
    torch.set_num_threads(min(16, torch.get_num_threads()))
    cfg = ModelConfig(timesteps=1000, noise_precision=0.05, norm_values=(1.0, 0.5))
    sd = make_state_dict(cfg, seed=5, coord_gain=1.0)
    p = ref_cpu.to_torch_params(sd)
    K = 1000
    for first in (10800, 8000, 7900, 10900):        # seeds whose oracle chain keeps >= 2e-5 A from the cutoff (found offline; re-checked below)
        pb = make_pockets(2, 'CA', n_phar=9, first_index=first)
        nl = int(pb.num_nodes_phar.sum())
        noise = torch.randn((K + 2, nl, 11), generator=torch.Generator().manual_seed(first))
        margins, orig = [], ref_cpu.get_edges

        def watched(mask, x, cutoff):
            margins.append(min_cutoff_margin(x.numpy(), mask.numpy(), cutoff))
            return orig(mask, x, cutoff)
        ref_cpu.get_edges = watched
        try:
            tape = iter(noise)
            with torch.no_grad():
                want, want_p, _, _ = ref_cpu.sample_given_pocket(p, cfg.as_dict(), pocket_dict(pb), pb.num_nodes_phar,
                                                                 timesteps=K, noise=lambda shape: next(tape))
        finally:
            ref_cpu.get_edges = orig
        if min(margins) < 2e-5:
            continue
        h = new_handle(cfg, sd)
        h.set_layout(pb.num_nodes_phar, pb.size)
        h.set_step_table(K, host_step_table(cfg, K))
        got, got_p, _ = h.sample_chain(dev(pb.x), dev(pb.one_hot), K, noise=noise.cuda())
        want = want.numpy()
        err = rms(got[:, :3].cpu().numpy(), want[:, :3])
        print(f'T=K=1000 bounded: RMS vs oracle {err:.3e} A (max|x| {np.abs(want[:, :3]).max():.1f} A, margin {min(margins):.1e})')
        assert np.abs(want[:, :3]).max() < 40.0
        assert err <= 1e-4
        assert np.array_equal(got[:, 3:].cpu().numpy(), want[:, 3:])
        assert rms(got_p[:, :3].cpu().numpy(), want_p[:, :3].numpy()) <= 1e-4
        h.close()
        return
    pytest.skip('every candidate seed came within 2e-5 A of the cutoff')


# ----------------------------------------------------------------------------- G5: per-block intermediates
def test_per_block_intermediates_match_reference():
    """The fused kernels never store m_ij / e_ij / trans; what they do produce per block is compared with the
    reference's intermediates: agg = sum_j e_ij / 100 (egnn_new.py:50-52) after the edge-message kernel, h after the
    node kernel (:56-57), the coordinate aggregate sum_j trans / 100 (:96-98, phar rows) after the coordinate kernel,
    and x after the block - for EVERY block (cmdgen_debug_eval_prefix)."""
    cfg, sd, inp = g5_case(G5)
    h = new_handle(cfg, sd)
    h.set_layout(G5['num_nodes_phar'], G5['pocket_size'])
    xp, xq, t = dev(inp['xh_phar']), dev(inp['xh_pocket']), dev(inp['t'])
    N, nl, H, L = len(inp['mask_phar']) + len(inp['mask_pocket']), len(inp['mask_phar']), cfg.hidden_nf, cfg.n_layers
    for b in range(L):
        h.debug_eval_prefix(xp, xq, t, b, 1)
        agg = h.debug_read('agg', N * H).reshape(N, H) / cfg.normalization_factor
        want = G5[f'block{b}/agg']
        assert np.abs(agg - want).max() <= EVAL_TOL * max(1.0, np.abs(want).max()), ('agg', b)
        h.debug_eval_prefix(xp, xq, t, b, 2)
        hb = h.debug_read('h', N * H).reshape(N, H)
        want = G5[f'block{b}/h']
        assert np.abs(hb - want).max() <= EVAL_TOL * max(1.0, np.abs(want).max()), ('h', b)
        assert not h.debug_read('agg', N * H).any()                       # the node kernel zeroes what it read
        h.debug_eval_prefix(xp, xq, t, b, 3)
        acc = h.debug_read('acc', L * nl * 4).reshape(L, nl, 4)[b, :, :3] / cfg.normalization_factor
        want = G5[f'block{b}/x_agg'][:nl]                                  # pocket rows are masked out (:100-101)
        assert np.abs(acc - want).max() <= EVAL_TOL * max(1.0, np.abs(want).max()), ('x_agg', b)
        x0 = h.debug_read('x0', nl * 4).reshape(nl, 4)[:, :3]
        xl = h.debug_read('xl', L * nl * 4).reshape(L, nl, 4)[:, :, :3]
        x_in = x0 if b == 0 else xl[b]
        want_x = G5[f'block{b}/x'][:nl]
        assert np.abs(x_in + acc - want_x).max() <= EVAL_TOL * max(1.0, np.abs(want_x).max()), ('x', b)
    eps, _ = h.dynamics_forward(xp, xq, t)                                # the workspace recovers after prefix runs
    assert np.abs(eps.cpu().numpy() - G5['eps_phar']).max() <= EVAL_TOL * max(1.0, np.abs(G5['eps_phar']).max())
    assert np.array_equal(h.get_edges(), G5['edges'])
    h.close()


# ----------------------------------------------------------------------------- G7: generate_phars end to end
from test_oracle_golden_r2 import _hparams, G7_RUNS, g7_select, assert_same_result  # noqa: E402


@pytest.mark.parametrize('tag,rep,sel', G7_RUNS)
def test_generate_phars_matches_reference_output(tag, rep, sel):
    """PharPocketDDPM.generate_phars on the GPU path (PDB reader -> pocket tensors -> HIP chain with the reference's
    recorded draws -> COM restore -> dict) against the dict the REFERENCE returned for the same structure."""
    from cmdgen_amd.lightning_modules import PharPocketDDPM
    key = 'ca' if rep == 'CA' else 'fa'
    H, L, R, seed = [int(v) for v in G7[key + '/meta']]
    model = PharPocketDDPM(**_hparams(rep, H, L))
    sd = make_state_dict(ModelConfig(hidden_nf=H, n_layers=L, residue_nf=R, timesteps=500), seed=seed, coord_gain=1e-3)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model = model.cuda()
    nph = G7[tag + '/num_nodes_phar']
    out = model.generate_phars(os.path.join(GOLDEN, 'g7_pocket.pdb'), len(nph), num_nodes_phar=torch.from_numpy(nph),
                               timesteps=int(G7[tag + '/K']), noise=dev(G7[tag + '/noise']), **g7_select(G7, tag, sel))
    want = json.loads(str(G7[tag + '/result_json']))
    scale = max(1.0, max(abs(v) for m in want.values() for cs in m.values() for c in cs for v in c))
    assert_same_result(out, str(G7[tag + '/result_json']), atol=1e-4 * scale)


# ----------------------------------------------------------------------------- get_edges(mask, x)
def test_get_edges_computes_the_graph_of_its_arguments():
    """EGNNDynamics.get_edges(batch_mask, x) (dynamics.py:141-147) for arbitrary input: the reference's boundary
    fixture (a pair at exactly 6.0 A, self loops), an interleaved (non-ascending) mask, and the phar-first
    concatenation forward() uses - each against the reference's / oracle's edge list, exact incl. order."""
    from oracle import ref_cpu
    from cmdgen_amd.equivariant_diffusion.dynamics import EGNNDynamics
    dyn = EGNNDynamics(phar_nf=8, residue_nf=20, n_dims=3, joint_nf=32, hidden_nf=64, n_layers=1, attention=True, tanh=True,
                       norm_constant=1, inv_sublayers=1, sin_embedding=False, normalization_factor=100,
                       aggregation_method='sum', edge_cutoff=6.0, update_pocket_coords=False).cuda()
    g3 = load_golden('g3_edges.npz')
    e = dyn.get_edges(dev(g3['mask']), dev(g3['x']))
    assert e.dtype == torch.int64 and np.array_equal(e.cpu().numpy(), g3['edges'])
    rng = np.random.Generator(np.random.PCG64(99))
    for trial in range(3):
        n = int(rng.integers(40, 400))
        x = rng.uniform(-9, 9, size=(n, 3)).astype(np.float32)
        mask = rng.integers(0, 5, size=n).astype(np.int64)                  # interleaved samples, some possibly empty
        if min_cutoff_margin(x, mask, 6.0) < 1e-4:
            continue
        row, col = ref_cpu.get_edges(torch.from_numpy(mask), torch.from_numpy(x), 6.0)
        got = dyn.get_edges(dev(mask), dev(x)).cpu().numpy()
        assert np.array_equal(got, np.stack([row.numpy(), col.numpy()])), trial
    # the layout of forward(): phar rows of all samples first, then pocket rows (golden from the reference)
    cfg, sd, inp = dynamics_case(G12, 'dyn_fa366_b2')
    mask = np.concatenate([inp['mask_phar'], inp['mask_pocket']])
    x = np.concatenate([inp['xh_phar'][:, :3], inp['xh_pocket'][:, :3]])
    got = dyn.get_edges(dev(mask), dev(x)).cpu().numpy()
    assert np.array_equal(got, G12['dyn_fa366_b2/edges'].astype(np.int64))
    # no cutoff: the complete graph per sample
    dyn.edge_cutoff = None
    dyn._cfg['edge_cutoff'] = None
    dyn._handle = None
    m = np.array([0, 0, 1, 1, 1], dtype=np.int64)
    got = dyn.get_edges(dev(m), dev(np.zeros((5, 3), np.float32))).cpu().numpy()
    assert got.shape[1] == 4 + 9


def test_large_sample_needs_the_lds_opt_in():
    """A 3000-node sample needs 72 KB of dynamic LDS in the neighbour search (above the 64 KiB default)."""
    from oracle import ref_cpu
    h = hip_backend.Handle(ModelConfig(hidden_nf=64, n_layers=1).as_dict(), 0)
    rng = np.random.Generator(np.random.PCG64(3))
    n = 3000
    x = (rng.uniform(-1, 1, size=(n, 3)) * 40).astype(np.float32)
    row, col = h.radius_graph(dev(x), [n])
    r, c = ref_cpu.get_edges(torch.zeros(n, dtype=torch.int64), torch.from_numpy(x), 6.0)
    if min_cutoff_margin(x, np.zeros(n, np.int64), 6.0) > 1e-5:
        assert np.array_equal(row.cpu().numpy(), r.numpy()) and np.array_equal(col.cpu().numpy(), c.numpy())
    else:
        assert abs(len(row) - len(r)) <= 4
    with pytest.raises(hip_backend.CmdgenError):
        h.set_layout([10], [7000])                                          # beyond the LDS bound: refused, not a launch failure
    h.close()


# ----------------------------------------------------------------------------- per-call noise
def test_consecutive_sampling_calls_draw_fresh_noise():
    """The reference draws new torch.randn noise on every call; so do we (a fresh Philox seed per call from torch's
    global generator) - and torch.manual_seed still makes a run reproducible."""
    from cmdgen_amd.equivariant_diffusion.dynamics import EGNNDynamics
    from cmdgen_amd.equivariant_diffusion.conditional_model import ConditionalDDPM
    cfg = ModelConfig(hidden_nf=64, n_layers=2, timesteps=100)
    sd = make_state_dict(cfg, seed=1)
    dyn = EGNNDynamics(phar_nf=8, residue_nf=20, n_dims=3, joint_nf=32, hidden_nf=64, n_layers=2, attention=True, tanh=True,
                       norm_constant=1, inv_sublayers=1, sin_embedding=False, normalization_factor=100,
                       aggregation_method='sum', edge_cutoff=6.0, update_pocket_coords=False)
    ddpm = ConditionalDDPM(dynamics=dyn, phar_nf=8, residue_nf=20, n_dims=3, timesteps=100, noise_schedule='polynomial_2',
                           noise_precision=1e-5, loss_type='l2', norm_values=[1, 4], size_histogram=np.ones((30, 70)))
    ddpm.load_state_dict({k[len('ddpm.'):]: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    ddpm = ddpm.cuda()
    pb = make_pockets(3, 'CA', n_phar=6)
    pocket = {k: v.cuda() for k, v in pocket_dict(pb).items()}
    nph = torch.from_numpy(pb.num_nodes_phar)
    torch.manual_seed(11)
    a = ddpm.sample_given_pocket(pocket, nph, timesteps=10)[0].cpu()
    b = ddpm.sample_given_pocket(pocket, nph, timesteps=10)[0].cpu()
    assert not torch.allclose(a[:, :3], b[:, :3])                           # second call: new noise
    torch.manual_seed(11)
    a2 = ddpm.sample_given_pocket(pocket, nph, timesteps=10)[0].cpu()
    assert torch.allclose(a, a2, atol=1e-3 * max(1.0, float(a.abs().max())))   # reproducible under manual_seed
    # identical pockets inside one call still get different noise (keyed by pocket id)
    assert not torch.allclose(a[:6, :3] - a[:6, :3].mean(0), a[6:12, :3] - a[6:12, :3].mean(0))


def test_bench_multi_rank_path_rehearsed_on_one_gpu():
    """bench.py --gpus 2 with both ranks on cuda:0 over gloo (--rehearse-on-one-gpu): self-launch, per-rank pocket shards keyed by
    global ids, barrier + MAX-over-ranks timing, rank count by all-reduce, ONE JSON line from rank 0 - the whole multi-rank path
    on the hardware a one-GPU box has.  The value is not a scaling result."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--rehearse-on-one-gpu', '--batch', '8',
                        '--timesteps', '40', '--steps', '1', '--warmup', '1', '--no-cpu-baseline'], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['value'] > 0 and out['scaling'] == 'weak' and 'rehearsal' in out
    assert out['config']['pockets_per_gpu'] == 8 and out['cpu_baseline'] is None
