"""GPU parity tests of the 128-row edge kernels (cmdgen_amd/csrc/kernels_edge128.hip: chunked edge lists, quarter-K plane builds, the
epilogue in registers, the segment sum by receiver as a matrix product).

The small fixtures pick smaller tiles by themselves, so these tests FORCE edge_mt = coord_mt = 128.  Same tolerances as everywhere
else: one evaluation max|d eps| <= 2e-5 * max(1, max|eps|) against the REFERENCE's output, chains <= 1e-4 A absolute coordinate RMS in
the bounded regime, types exact.
"""
import numpy as np
import pytest
import torch

from helpers import load_golden, cases_of, dynamics_case, rms, bounded_case
from cmdgen_amd import hip_backend
from cmdgen_amd.synthetic import ModelConfig, make_state_dict, make_pockets
from test_hip_parity_r2 import dev, new_handle, host_step_table, EVAL_TOL

pytestmark = pytest.mark.gpu

G2 = load_golden('g2_dynamics.npz')
G12 = load_golden('g12_fullsize.npz')
G13 = load_golden('g13_bounded.npz')


@pytest.fixture(autouse=True, params=[(2, 3), (2, 0), (0, 0)], ids=['half_fused', 'half_plain', 'bf16x3'])
def e128_engine(request, monkeypatch):
    """every test of this file runs on both matrix engines of the 128-row kernels - two fp16 pieces per operand and three MFMAs per product (the
    default where the model has an edge cutoff, forced here with half_engine = 2), three bf16 pieces and six (half_engine = 0) - and, on the half
    engine, on both main loops: the fused one (round 6: the next quarter's tile build inside the GEMM; what long lists run) and the plain one
    (what a list of one tile per workgroup runs); option e128_fused"""
    monkeypatch.setitem(hip_backend.DEFAULT_OPTIONS, 'half_engine', request.param[0])
    monkeypatch.setitem(hip_backend.DEFAULT_OPTIONS, 'e128_fused', request.param[1])
    return request.param


def force128(monkeypatch):
    """every Handle created from here on starts with these options (hip_backend.DEFAULT_OPTIONS -> cmdgen_set_option)"""
    monkeypatch.setitem(hip_backend.DEFAULT_OPTIONS, 'edge_mt', 128)
    monkeypatch.setitem(hip_backend.DEFAULT_OPTIONS, 'coord_mt', 128)


@pytest.mark.parametrize('name', [n for n in cases_of(G2) if '_h256_' in n] + ['G12:dyn_fa366_b2'])
def test_evaluation_matches_reference(name, monkeypatch):
    """Every H = 256 evaluation fixture (B == 1 scalar-time branch, ragged Nl, full-atom vocabulary, Np = 366) with both edge kernels on
    128-row tiles, against the reference's eps; twice on the same handle (agg left zero)."""
    force128(monkeypatch)
    g = G2
    if name.startswith('G12:'):
        g, name = G12, name[4:]
    cfg, sd, inp = dynamics_case(g, name)
    want = g[name + '/eps_phar']
    h = new_handle(cfg, sd)
    h.set_layout(g[name + '/num_nodes_phar'], g[name + '/pocket_size'])
    assert h.query('edge_mt') == 128 and h.query('coord_mt') == 128 and h.query('gemm_split') == 1
    outs = []
    for _ in range(2):
        eps, _p = h.dynamics_forward(dev(inp['xh_phar']), dev(inp['xh_pocket']), dev(inp['t']))
        torch.cuda.synchronize()
        outs.append(eps.cpu().numpy())
    err = float(np.abs(outs[0] - want).max())
    tol = EVAL_TOL * max(1.0, float(np.abs(want).max()))
    print(f'{name}: max|d eps| {err:.2e} (tolerance {tol:.1e}); run-to-run max diff {float(np.abs(outs[0] - outs[1]).max()):.1e}')
    assert err <= tol
    assert float(np.abs(outs[0] - outs[1]).max()) <= 1e-6 * max(1.0, float(np.abs(want).max()))     # (a receiver spread over three or more tiles takes three float atomics)
    h.close()


def test_per_block_intermediates_match_reference(monkeypatch):
    """agg after the message kernel, h after the node kernel, the coordinate sums after the coordinate kernel, for every block (G5)."""
    import test_hip_parity_r2
    force128(monkeypatch)
    test_hip_parity_r2.test_per_block_intermediates_match_reference()


@pytest.mark.parametrize('seed', range(8))
def test_fuzz_configurations(seed, monkeypatch):
    """Randomised configurations (flags on / off, complete graphs - segments of more than 128 edges and tiles of more than 32 segments -,
    ragged batches) against the oracle; hidden_nf != 256 falls back to the 64-row kernels."""
    import test_hip_parity
    force128(monkeypatch)
    test_hip_parity.test_fuzz_hyperparameters_and_layouts(seed)


def test_joint_model(monkeypatch):
    """Joint model: every receiver moves (the coordinate list is the whole list without self loops)."""
    import test_hip_joint
    force128(monkeypatch)
    test_hip_joint.test_joint_dynamics_fuzz_vs_oracle()


@pytest.mark.parametrize('use_graph', [False, True])
def test_bounded_chain_absolute_rms(use_graph, monkeypatch):
    """The reference's K = T = 500 chain (G13) with every edge launch on the 128-row kernels: <= 1e-4 A absolute, types exact."""
    force128(monkeypatch)
    name = [n for n in cases_of(G13)][-1]
    cfg, sd, pb, K = bounded_case(G13, name)
    h = new_handle(cfg, sd)
    h.set_layout(pb.num_nodes_phar, pb.size)
    assert h.query('edge_mt') == 128
    h.set_step_table(K, host_step_table(cfg, K))
    xh_phar, xh_pocket, _ = h.sample_chain(dev(pb.x), dev(pb.one_hot), K, noise=dev(G13[name + '/noise']), use_graph=use_graph)
    st = h.chain_status()
    want = G13[name + '/xh_phar']
    err = rms(xh_phar[:, :3].cpu().numpy(), want[:, :3])
    print(f'{name} graph={use_graph}: coordinate RMS vs reference {err:.3e} A')
    assert err <= 1e-4
    assert np.array_equal(xh_phar[:, 3:].cpu().numpy(), want[:, 3:])
    assert st['max_rel_com_error'] < 1e-2 and st['nan_resets'] == 0
    h.close()


def test_large_batch_agrees_with_the_64_row_kernels(monkeypatch):
    """256 C-alpha pockets at the geometry a trained model holds and 16 full-atom pockets: one evaluation on the 128-row kernels against the same
    evaluation on the 64-row kernels (which the fixtures above pin to the reference): max|d eps| <= 2e-5 * max(1, |eps|)."""
    for B, rep in ((256, 'CA'), (16, 'full-atom')):
        cfg = ModelConfig(residue_nf=20 if rep == 'CA' else 11, timesteps=1000)
        sd = make_state_dict(cfg, seed=0)
        pb = make_pockets(B, rep)
        rng = np.random.Generator(np.random.PCG64(7))
        nl = int(pb.num_nodes_phar.sum())
        pm = np.repeat(np.arange(B), pb.num_nodes_phar)
        com = np.stack([pb.x[pb.mask == b].mean(0) for b in range(B)])
        v = rng.normal(size=(nl, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
        xin = (com[pm] + v * 5.0 * np.cbrt(rng.uniform(size=(nl, 1)))).astype(np.float32)
        xh = dev(np.concatenate([xin, rng.normal(size=(nl, cfg.phar_nf)).astype(np.float32)], 1))
        xq = dev(np.concatenate([pb.x, pb.one_hot / cfg.norm_values[1]], 1).astype(np.float32))
        t = dev(np.full((B,), 0.5, np.float32))
        out = {}
        for mt in ('64', '128'):
            h = new_handle(cfg, sd)
            h.set_option('edge_mt', int(mt)); h.set_option('coord_mt', int(mt))
            h.set_layout(pb.num_nodes_phar, pb.size)
            assert h.query('edge_mt') == int(mt)
            eps, _ = h.dynamics_forward(xh, xq, t)
            torch.cuda.synchronize()
            out[mt] = eps.cpu().numpy()
            h.close()
        d = float(np.abs(out['64'] - out['128']).max())
        sc = max(1.0, float(np.abs(out['64']).max()))
        print(f'{B} {rep} pockets: max|eps_128 - eps_64| {d:.2e} (|eps| max {sc:.2f})')
        assert np.isfinite(out['128']).all() and d <= EVAL_TOL * sc


def test_dead_work_skip_with_128_row_tiles(monkeypatch):
    """A drifted 100-step chain of 64 pockets with the dead-tile skip on and off: same result, tiles skipped only when on."""
    force128(monkeypatch)
    cfg = ModelConfig(residue_nf=20, timesteps=1000)
    sd = make_state_dict(cfg, seed=0)
    pb = make_pockets(64, 'CA')
    out = {}
    for flag in ('2', '0'):
        monkeypatch.setitem(hip_backend.DEFAULT_OPTIONS, 'dead_skip', int(flag))
        h = new_handle(cfg, sd)
        h.set_layout(pb.num_nodes_phar, pb.size)
        h.reset_counters()
        xh, _, _ = h.sample_chain(dev(pb.x), dev(pb.one_hot), 100, seed=3)
        out[flag] = (xh.cpu().numpy(), h.counters())
        assert h.chain_status()['nan_resets'] == 0
        h.close()
    on, off = out['2'][1], out['0'][1]
    assert off['edges_skipped'] == 0 and on['edges_skipped'] > 0
    sc = max(1.0, float(np.abs(out['0'][0][:, :3]).max()))
    assert np.abs(out['2'][0][:, :3] - out['0'][0][:, :3]).max() <= 2e-5 * sc and np.array_equal(out['2'][0][:, 3:], out['0'][0][:, 3:])


def test_nonfinite_message_stays_with_its_receiver_and_resets_the_batch(monkeypatch):
    """A pocket node of ONE sample carries an infinite feature: its messages are non-finite, the velocity has a NaN and the evaluation resets
    EVERY velocity of the batch to zero like the reference (dynamics.py:129-131, batch-global).  What must not happen on the 128-row tiles -
    where receivers of several samples share a tile - is the poison reaching another sample's receivers through the segment sum: the decoded
    features of every other sample still equal the oracle's (the ordered in-register scan keeps a NaN in its own receiver's sum; the
    segment-sum-as-MFMA of round 4 spread it over the tile through 0 * NaN)."""
    from oracle import ref_cpu
    force128(monkeypatch)
    name = 'ca_h256_b8'
    cfg, sd, inp = dynamics_case(G2, name)
    nl, npk = G2[name + '/num_nodes_phar'], G2[name + '/pocket_size']
    bad_sample = 3
    xq = inp['xh_pocket'].copy()
    xq[int(np.cumsum(npk)[bad_sample - 1]) + 2, 3 + 1] = np.inf             # one feature of the third pocket node of sample 3
    p = ref_cpu.to_torch_params(sd)
    with torch.no_grad():
        want, _ = ref_cpu.dynamics_forward(p, cfg.as_dict(), torch.from_numpy(inp['xh_phar']), torch.from_numpy(xq), torch.from_numpy(inp['t']),
                                           torch.from_numpy(inp['mask_phar']), torch.from_numpy(inp['mask_pocket']))
    want = want.numpy()
    h = new_handle(cfg, sd)
    h.set_layout(nl, npk)
    assert h.query('edge_mt') == 128
    eps, _p = h.dynamics_forward(dev(inp['xh_phar']), dev(xq), dev(inp['t']))
    torch.cuda.synchronize()
    got = eps.cpu().numpy()
    assert np.all(want[:, :3] == 0.0) and np.all(got[:, :3] == 0.0)        # the batch-global reset, in the oracle and here
    others = inp['mask_phar'] != bad_sample
    assert np.isfinite(want[others, 3:]).all()
    assert np.isfinite(got[others, 3:]).all(), 'a non-finite message leaked into another sample'
    assert np.abs(got[others, 3:] - want[others, 3:]).max() <= EVAL_TOL * max(1.0, float(np.abs(want[others, 3:]).max()))
    assert not np.isfinite(got[~others, 3:]).all()                          # the poisoned sample itself is non-finite in both
    assert not np.isfinite(want[~others, 3:]).all()
    h.close()
