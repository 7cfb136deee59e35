"""GPU parity tests of the split-bf16 matrix engine (cmdgen_amd/csrc/cmdgen_split.h, cmdgen_set_gemm_mode).

Tiles of >= 32 rows multiply every fp32 product as six exact bf16 products accumulated in fp32; 16-row tiles keep
v_mfma_f32_16x16x4_f32.  Small fixtures pick 16-row tiles by themselves, so these tests FORCE 32- and 64-row tiles
(handle options node_mt / edge_mt / coord_mt, cmdgen_set_option) to put all three MFMA kernels and k_embed on the split engine, and run the same case on the fp32
engine next to it.  Same tolerances as everywhere else: one evaluation max|d eps| <= 2e-5 * max(1, max|eps|) against the
reference's output, chains <= 1e-4 A ABSOLUTE coordinate RMS in the bounded regime, types exact.
"""
import numpy as np
import pytest
import torch

from helpers import load_golden, cases_of, dynamics_case, rms, bounded_case, fullsize_chain_case
from cmdgen_amd import hip_backend
from test_hip_parity_r2 import dev, new_handle, host_step_table, EVAL_TOL

pytestmark = pytest.mark.gpu

G2 = load_golden('g2_dynamics.npz')
G12 = load_golden('g12_fullsize.npz')
G13 = load_golden('g13_bounded.npz')


def force_tiles(monkeypatch, mt):
    """every Handle created from here on starts with these options (hip_backend.DEFAULT_OPTIONS -> cmdgen_set_option)"""
    for k in ('node_mt', 'edge_mt', 'coord_mt'):
        monkeypatch.setitem(hip_backend.DEFAULT_OPTIONS, k, int(mt))


@pytest.mark.parametrize('mt', [32, 64])
@pytest.mark.parametrize('name', [n for n in cases_of(G2) if '_h32_' not in n])   # hidden_nf 32 is below the 64-column wave tile
def test_evaluation_on_both_engines_matches_reference(name, mt, monkeypatch):
    """Every G2 evaluation fixture (H in {64,128,256}, flags on/off, ragged) with all tiles forced to mt rows."""
    force_tiles(monkeypatch, mt)
    cfg, sd, inp = dynamics_case(G2, name)
    want = G2[name + '/eps_phar']
    errs = {}
    for split in (True, False):
        h = new_handle(cfg, sd)
        h.set_gemm_mode(split)
        h.set_layout(G2[name + '/num_nodes_phar'], G2[name + '/pocket_size'])
        assert h.query('gemm_split') == int(split) and h.query('edge_mt') == mt
        eps, _ = h.dynamics_forward(dev(inp['xh_phar']), dev(inp['xh_pocket']), dev(inp['t']))
        torch.cuda.synchronize()
        errs[split] = float(np.abs(eps.cpu().numpy() - want).max())
        h.close()
    tol = EVAL_TOL * max(1.0, float(np.abs(want).max()))
    print(f'{name} mt={mt}: max|d eps| split {errs[True]:.2e}  fp32 {errs[False]:.2e}  (tolerance {tol:.1e})')
    assert errs[True] <= tol and errs[False] <= tol


def test_fullsize_evaluation_split_error_is_at_the_fp32_engines_level():
    """configs[4]'s shape (Np=366): the split engine's deviation from the reference is of the size of the fp32 engine's."""
    name = 'dyn_fa366_b2'
    cfg, sd, inp = dynamics_case(G12, name)
    want = G12[name + '/eps_phar']
    errs = {}
    for split in (True, False):
        h = new_handle(cfg, sd)
        h.set_gemm_mode(split)
        h.set_layout(G12[name + '/num_nodes_phar'], G12[name + '/pocket_size'])
        eps, _ = h.dynamics_forward(dev(inp['xh_phar']), dev(inp['xh_pocket']), dev(inp['t']))
        torch.cuda.synchronize()
        errs[split] = rms(eps.cpu().numpy(), want)
        h.close()
    print(f'RMS deviation from the reference: split {errs[True]:.3e}, fp32 {errs[False]:.3e}')
    assert errs[True] <= 3.0 * errs[False] + 1e-7


@pytest.mark.parametrize('use_graph', [False, True])
@pytest.mark.parametrize('name', cases_of(G13))
def test_bounded_chain_on_split_engine_absolute_rms(name, use_graph, monkeypatch):
    """The absolute 1e-4 A bound with EVERY tile kernel of every step on the split engine (32-row tiles forced):
    K = 50 and the full K = T = 500 reference chains, recorded noise."""
    force_tiles(monkeypatch, 32)
    cfg, sd, pb, K = bounded_case(G13, name)
    h = new_handle(cfg, sd)
    h.set_layout(pb.num_nodes_phar, pb.size)
    assert h.query('gemm_split') == 1 and h.query('node_mt') == 32 and h.query('edge_mt') == 32
    h.set_step_table(K, host_step_table(cfg, K))
    xh_phar, xh_pocket, _ = h.sample_chain(dev(pb.x), dev(pb.one_hot), K, noise=dev(G13[name + '/noise']), use_graph=use_graph)
    st = h.chain_status()
    want = G13[name + '/xh_phar']
    err = rms(xh_phar[:, :3].cpu().numpy(), want[:, :3])
    print(f'{name} graph={use_graph} split engine: coordinate RMS vs reference {err:.3e} A')
    assert err <= 1e-4
    assert np.array_equal(xh_phar[:, 3:].cpu().numpy(), want[:, 3:])
    assert st['max_rel_com_error'] < 1e-2 and st['nan_resets'] == 0
    h.close()


def test_mode_switch_recaptures_the_step_graph(monkeypatch):
    """cmdgen_set_gemm_mode between two graph-replayed chains of one handle: both match the reference chain."""
    force_tiles(monkeypatch, 32)
    name = cases_of(G13)[0]
    cfg, sd, pb, K = bounded_case(G13, name)
    h = new_handle(cfg, sd)
    h.set_layout(pb.num_nodes_phar, pb.size)
    h.set_step_table(K, host_step_table(cfg, K))
    want = G13[name + '/xh_phar']
    outs = []
    for split in (True, False, True):
        h.set_gemm_mode(split)
        xh_phar, _, _ = h.sample_chain(dev(pb.x), dev(pb.one_hot), K, noise=dev(G13[name + '/noise']), use_graph=True)
        assert rms(xh_phar[:, :3].cpu().numpy(), want[:, :3]) <= 1e-4
        outs.append(xh_phar.cpu().numpy())
    assert np.array_equal(outs[0][:, 3:], outs[1][:, 3:])
    assert np.allclose(outs[0], outs[2], atol=1e-5)          # same engine, same chain (up to float atomics at tile seams)
    h.close()


@pytest.mark.parametrize('mt', [32, 64])
@pytest.mark.parametrize('seed', range(8))
def test_fuzz_configurations_on_split_engine(seed, mt, monkeypatch):
    """The randomised configurations of test_hip_parity (hidden_nf 64 / 128 / 256, 1-5 blocks, flags on / off, complete graphs,
    no time conditioning, other vocabulary sizes, ragged batches) against the oracle with every tile kernel forced onto the
    split engine - in particular the short k loops of hidden_nf 64 (K/16 = 4) and 128."""
    import test_hip_parity
    force_tiles(monkeypatch, mt)
    test_hip_parity.test_fuzz_hyperparameters_and_layouts(seed)


@pytest.mark.parametrize('seed', range(4))
def test_fuzz_chains_on_split_engine(seed, monkeypatch):
    """Randomised short chains (hidden_nf 64 / 128, with and without the COM projection, per-step states) vs the oracle chain,
    32-row tiles forced: every evaluation of the chain on the split engine."""
    import test_hip_parity
    force_tiles(monkeypatch, 32)
    test_hip_parity.test_fuzz_chains(seed)


@pytest.mark.parametrize('mt', [32, 64])
def test_joint_model_on_split_engine(mt, monkeypatch):
    """The joint model's evaluation (every receiver moves, velocity COM removed): the ragged hidden_nf 64 / 128 / 256 fuzz of
    test_hip_joint against the oracle with every tile forced onto the split engine."""
    import test_hip_joint
    force_tiles(monkeypatch, mt)
    test_hip_joint.test_joint_dynamics_fuzz_vs_oracle()


# ----------------------------------------------------------------------------- 16-row tiles on the split engine (round 3)
@pytest.mark.parametrize('name', [n for n in cases_of(G2) if '_h256_' in n or '_h128_' in n])
def test_node_kernel_16_row_split_tiles_match_reference(name, monkeypatch):
    """k_node<H, 16> on v_mfma_f32_16x16x32_bf16 (Eng<16, true>, option node16_split = 1): every G2 evaluation fixture with
    H >= 128 against the reference's output, next to the fp32 16-row kernel."""
    monkeypatch.setitem(hip_backend.DEFAULT_OPTIONS, 'node_mt', 16)
    cfg, sd, inp = dynamics_case(G2, name)
    want = G2[name + '/eps_phar']
    errs = {}
    for s16 in ('1', '0'):
        monkeypatch.setitem(hip_backend.DEFAULT_OPTIONS, 'node16_split', int(s16))
        h = new_handle(cfg, sd)
        h.set_layout(G2[name + '/num_nodes_phar'], G2[name + '/pocket_size'])
        assert h.query('node_mt') == 16
        eps, _ = h.dynamics_forward(dev(inp['xh_phar']), dev(inp['xh_pocket']), dev(inp['t']))
        torch.cuda.synchronize()
        errs[s16] = float(np.abs(eps.cpu().numpy() - want).max())
        h.close()
    tol = EVAL_TOL * max(1.0, float(np.abs(want).max()))
    print(f'{name} node tiles of 16 rows: max|d eps| split {errs["1"]:.2e}  fp32 {errs["0"]:.2e}  (tolerance {tol:.1e})')
    assert errs['1'] <= tol and errs['0'] <= tol


@pytest.mark.parametrize('use_graph', [False, True])
@pytest.mark.parametrize('name', cases_of(G13))
def test_bounded_chain_with_16_row_split_node_tiles(name, use_graph, monkeypatch):
    """The G13 reference chains (K = 50, K = T = 500) with the node kernel on 16-row split tiles: 1e-4 A absolute."""
    monkeypatch.setitem(hip_backend.DEFAULT_OPTIONS, 'node_mt', 16)
    monkeypatch.setitem(hip_backend.DEFAULT_OPTIONS, 'node16_split', 1)
    cfg, sd, pb, K = bounded_case(G13, name)
    h = new_handle(cfg, sd)
    h.set_layout(pb.num_nodes_phar, pb.size)
    h.set_step_table(K, host_step_table(cfg, K))
    xh_phar, xh_pocket, _ = h.sample_chain(dev(pb.x), dev(pb.one_hot), K, noise=dev(G13[name + '/noise']), use_graph=use_graph)
    st = h.chain_status()
    want = G13[name + '/xh_phar']
    err = rms(xh_phar[:, :3].cpu().numpy(), want[:, :3])
    print(f'{name} graph={use_graph} 16-row split node tiles: coordinate RMS vs reference {err:.3e} A')
    assert err <= 1e-4
    assert np.array_equal(xh_phar[:, 3:].cpu().numpy(), want[:, 3:])
    assert st['max_rel_com_error'] < 1e-2 and st['nan_resets'] == 0
    h.close()


# ----------------------------------------------------------------------------- k_node64 (round 3): 64-row node tiles, both images in LDS
@pytest.mark.parametrize('name', [n for n in cases_of(G2) if '_h256_' in n] + ['dyn_fa366_b2'])
def test_node64_kernel_matches_reference(name, monkeypatch):
    """k_node64 (kernels_node64.hip) forced on (option node64 = 1) against the reference's output: every H = 256 evaluation fixture
    (ragged tiles, tiles that mix phar and pocket rows) and configs[4]'s shape (762 rows = 12 tiles), next to the 32-row kernel."""
    G = G12 if name.startswith('dyn_fa') else G2
    cfg, sd, inp = dynamics_case(G, name)
    want = G[name + '/eps_phar']
    errs = {}
    for n64 in ('1', '8', '2', '0'):      # 8: the 64-row tile on eight waves (k_node64e); 2: the lean 64-row tile, two workgroups per CU (k_node64d)
        monkeypatch.setitem(hip_backend.DEFAULT_OPTIONS, 'node64', int(n64))
        h = new_handle(cfg, sd)
        h.set_layout(G[name + '/num_nodes_phar'], G[name + '/pocket_size'])
        assert h.query('node64') == int(n64)
        eps, _ = h.dynamics_forward(dev(inp['xh_phar']), dev(inp['xh_pocket']), dev(inp['t']))
        eps, _ = h.dynamics_forward(dev(inp['xh_phar']), dev(inp['xh_pocket']), dev(inp['t']))      # agg must have been left zero
        torch.cuda.synchronize()
        errs[n64] = float(np.abs(eps.cpu().numpy() - want).max())
        h.close()
    tol = EVAL_TOL * max(1.0, float(np.abs(want).max()))
    print(f'{name}: max|d eps| 64-row kernel {errs["1"]:.2e}  eight-wave 64-row kernel {errs["8"]:.2e}  lean 64-row kernel {errs["2"]:.2e}  default {errs["0"]:.2e}  (tolerance {tol:.1e})')
    assert errs['1'] <= tol and errs['8'] <= tol and errs['2'] <= tol and errs['0'] <= tol


@pytest.mark.parametrize('use_graph', [False, True])
def test_bounded_chain_with_node64(use_graph, monkeypatch):
    """The K = T = 500 reference chain of G13 with the node kernel on 64-row tiles: 1e-4 A absolute, types exact."""
    monkeypatch.setitem(hip_backend.DEFAULT_OPTIONS, 'node64', 1)
    name = 'ca_h256_KT_np05'
    cfg, sd, pb, K = bounded_case(G13, name)
    h = new_handle(cfg, sd)
    h.set_layout(pb.num_nodes_phar, pb.size)
    assert h.query('node64') == 1
    h.set_step_table(K, host_step_table(cfg, K))
    xh_phar, xh_pocket, _ = h.sample_chain(dev(pb.x), dev(pb.one_hot), K, noise=dev(G13[name + '/noise']), use_graph=use_graph)
    st = h.chain_status()
    want = G13[name + '/xh_phar']
    err = rms(xh_phar[:, :3].cpu().numpy(), want[:, :3])
    print(f'{name} graph={use_graph} 64-row node tiles: coordinate RMS vs reference {err:.3e} A')
    assert err <= 1e-4 and np.array_equal(xh_phar[:, 3:].cpu().numpy(), want[:, 3:])
    assert st['max_rel_com_error'] < 1e-2 and st['nan_resets'] == 0
    h.close()


def test_node64_is_chosen_by_tile_count_and_agrees_with_the_32_row_kernel(monkeypatch):
    """Without the option node64 the launcher picks the plane node tiles where the eight-wave 16-row tile does not apply (round 6: the 32-ROW plane
    tile, two workgroups per CU - 128 C-alpha pockets; the 64-row tile on eight waves, k_node64e, where 64-row tiles fit one per CU and 32-row
    tiles do not - 256 pockets; the lean 64-row tile, two workgroups per CU, k_node64d, with more 64-row tiles than CUs - 384 pockets; not at the
    headline size: 64 pockets run k_node16w); 20-step chains of 256 pockets on the eight-wave 64-row tile, the 32-row plane tile, the lean and the
    full 64-row plane tile (options node64 = 2 / 1) and the register-split 32-row kernel (node64 = 0) agree to the engines' rounding."""
    from cmdgen_amd.synthetic import ModelConfig, make_state_dict, make_pockets
    monkeypatch.delitem(hip_backend.DEFAULT_OPTIONS, 'node64', raising=False)
    cfg = ModelConfig(residue_nf=20, timesteps=1000, noise_precision=0.1, norm_values=(1.0, 0.25))
    sd = make_state_dict(cfg, seed=3)
    out = {}
    out384 = {}
    for B, opt, expect in ((64, None, 0), (128, None, 32), (384, None, 2), (384, 32, 32), (256, None, 8), (256, 32, 32), (256, 2, 2), (256, 1, 1), (256, 0, 0)):
        pb = make_pockets(B, 'CA')
        h = hip_backend.Handle(cfg.as_dict(), 0); h.load_state_dict(sd)
        h.set_option('node64', opt)                                  # None: the library's own choice
        h.set_layout(pb.num_nodes_phar, pb.size)
        if torch.cuda.get_device_properties(0).multi_processor_count == 256: assert h.query('node64') == expect
        if B == 256:
            xh, _, _ = h.sample_chain(dev(pb.x), dev(pb.one_hot), 20, seed=5)
            out[h.query('node64')] = xh.cpu().numpy()
            assert h.chain_status()['nan_resets'] == 0
        if B == 384:         # the lean tile where it is the rule: two workgroups per CU
            xh, _, _ = h.sample_chain(dev(pb.x), dev(pb.one_hot), 20, seed=5)
            out384[h.query('node64')] = xh.cpu().numpy()
            assert h.chain_status()['nan_resets'] == 0
        h.close()
    if 2 in out384 and 32 in out384:
        err = rms(out384[2][:, :3], out384[32][:, :3])
        print(f'384 pockets, 20 steps: coordinate RMS lean 64-row plane tile (two per CU) vs 32-row plane tile {err:.2e} A')
        assert err <= 2e-5 and np.array_equal(out384[2][:, 3:], out384[32][:, 3:])
    tile_name = {1: '64-row', 32: '32-row', 8: 'eight-wave 64-row', 2: 'lean 64-row'}
    for k in (1, 32, 8, 2):
        if k in out and 0 in out:
            err = rms(out[k][:, :3], out[0][:, :3])
            print(f'256 pockets, 20 steps: coordinate RMS {tile_name[k]} plane tile vs register-split 32-row node kernel {err:.2e} A (max|x| {np.abs(out[0][:, :3]).max():.1f})')
            assert err <= 2e-5 and np.array_equal(out[k][:, 3:], out[0][:, 3:])
