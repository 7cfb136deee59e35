"""The half matrix engine's range (round 6; cmdgen_split.h: two fp16 pieces per operand).

An activation beyond fp16's 65504 becomes Inf, its second piece -Inf, the product NaN - and the evaluation's NaN guard turns that into a
batch-global reset step which the fp32 reference (whose activations stay finite) never takes.  These tests DRIVE such activations through each of
the half-engine tile kernels - k_edge128<msg>, k_edge128<coord>, the 32-row full-K edge tiles, k_node64, k_node16w - with weight sets whose
oracle output is finite, and check the library's contract:

  * the raw C-ABI call on a half-engine handle does reset (the overflow is real: the test reaches the path it is about) and reports it
    (cmdgen_counters.nan_resets / cmdgen_chain_status);
  * the Python mirror (hip_backend.Handle.run_range_guarded, used by EGNNDynamics.forward and by every chain entry point) never returns that
    result: it repeats the call on the three-piece bf16 split engine (fp32's exponent range), warns, and the output equals the ORACLE's within
    the evaluation tolerance.
"""
import warnings

import numpy as np
import pytest
import torch

from helpers import load_golden, dynamics_case
from cmdgen_amd import hip_backend
from test_hip_parity_r2 import dev, new_handle, EVAL_TOL, host_step_table

pytestmark = pytest.mark.gpu

G2 = load_golden('g2_dynamics.npz')
NAME = 'ca_h256_b8'
FACTOR = 3.0e6           # first-layer gain: the hidden activation of the targeted MLP reaches ~1e5 ... 1e6 (> 65504, far below fp32's 3e38)

# which first layer is scaled -> which kernel's A operand (the SiLU output feeding the second layer) overflows first
TARGETS = {
    'msg': 'ddpm.dynamics.egnn.e_block_1.gcl_0.edge_mlp.0',            # GCL.edge_model           (egnn_new.py:31-42)
    'coord': 'ddpm.dynamics.egnn.e_block_1.gcl_equiv.coord_mlp.0',     # EquivariantUpdate        (egnn_new.py:87-96)
    'node': 'ddpm.dynamics.egnn.e_block_1.gcl_0.node_mlp.0',           # GCL.node_model           (egnn_new.py:48-58)
}
# launch choices that put the three tile kernels on each half-engine family
OPTION_SETS = {
    'rows128_node64': dict(edge_mt=128, coord_mt=128, node64=1, e128_fused=0),
    'rows128_fused': dict(edge_mt=128, coord_mt=128, node64=1, e128_fused=3),
    'fullk32_node16w': dict(edge_mt=32, coord_mt=32, node_mt=16),
}


def overflow_case(target):
    cfg, sd, inp = dynamics_case(G2, NAME)
    sd = dict(sd)
    for suffix in ('.weight', '.bias'):
        sd[TARGETS[target] + suffix] = (sd[TARGETS[target] + suffix] * FACTOR).astype(np.float32)
    return cfg, sd, inp


def oracle_eps(cfg, sd, inp):
    from oracle import ref_cpu
    p = ref_cpu.to_torch_params(sd)
    with torch.no_grad():
        want, _ = ref_cpu.dynamics_forward(p, cfg.as_dict(), torch.from_numpy(inp['xh_phar']), torch.from_numpy(inp['xh_pocket']), torch.from_numpy(inp['t']),
                                           torch.from_numpy(inp['mask_phar']), torch.from_numpy(inp['mask_pocket']))
    return want.numpy()


@pytest.mark.parametrize('optset', list(OPTION_SETS))
@pytest.mark.parametrize('target', list(TARGETS))
def test_overflowing_activation_is_rerun_not_reset(target, optset, monkeypatch):
    for k, v in OPTION_SETS[optset].items():
        monkeypatch.setitem(hip_backend.DEFAULT_OPTIONS, k, v)
    cfg, sd, inp = overflow_case(target)
    want = oracle_eps(cfg, sd, inp)
    assert np.isfinite(want).all() and np.abs(want[:, :3]).max() > 0, 'the oracle must stay finite and take no reset'
    nl, npk = G2[NAME + '/num_nodes_phar'], G2[NAME + '/pocket_size']
    h = new_handle(cfg, sd)
    h.set_layout(nl, npk)
    assert h.half_engine_active()
    assert h.query({'msg': 'msg_mfmas_per_product', 'coord': 'coord_mfmas_per_product', 'node': 'node_mfmas_per_product'}[target]) == 3, \
        'the targeted kernel must run on the half engine for this test to mean anything'
    xp, xq, t = dev(inp['xh_phar']), dev(inp['xh_pocket']), dev(inp['t'])
    # 1. the raw C-ABI call: the half engine overflows, the guard resets the batch and counts it
    h.reset_counters()
    eps, _ = h.dynamics_forward(xp, xq, t)
    torch.cuda.synchronize()
    assert h.counters()['nan_resets'] == 1, 'expected the fp16 overflow to surface as a NaN reset on the raw half-engine call'
    assert np.all(eps.cpu().numpy()[:, :3] == 0.0)
    # 2. the mirror's guard: re-run on the bf16 split engine, warning, oracle-equal output
    seen = [h.nan_resets_total()]

    def status():
        now = h.nan_resets_total()
        d, seen[0] = now - seen[0], now
        return {'nan_resets': d}
    with pytest.warns(RuntimeWarning, match='half matrix engine'):
        (eps2, _p), st = h.run_range_guarded(lambda: h.dynamics_forward(xp, xq, t), status)
    torch.cuda.synchronize()
    got = eps2.cpu().numpy()
    assert st.get('half_engine_fallback') and st['nan_resets'] == 0
    assert h.half_engine_active(), 'the handle goes back to its own engine choice after the guarded call'
    tol = EVAL_TOL * max(1.0, float(np.abs(want).max()))
    err = float(np.abs(got - want).max())
    print(f'{target} / {optset}: |eps| max {np.abs(want).max():.3e}, guarded call vs oracle {err:.2e} (tolerance {tol:.1e})')
    assert np.isfinite(got).all() and err <= tol
    h.close()


def test_mirror_forward_equals_the_oracle_on_an_overflowing_model():
    """EGNNDynamics.forward (the reference's module interface) on a model whose message MLP overflows fp16: oracle-equal output, no reset."""
    from cmdgen_amd.equivariant_diffusion.dynamics import EGNNDynamics
    cfg, sd, inp = overflow_case('msg')
    want = oracle_eps(cfg, sd, inp)
    c = cfg.as_dict()
    dyn = EGNNDynamics(phar_nf=c['phar_nf'], residue_nf=c['residue_nf'], n_dims=3, joint_nf=c['joint_nf'], hidden_nf=c['hidden_nf'],
                       n_layers=c['n_layers'], attention=c['attention'], tanh=c['tanh'], norm_constant=c['norm_constant'],
                       inv_sublayers=c.get('inv_sublayers', 1), sin_embedding=False, normalization_factor=c['normalization_factor'],
                       aggregation_method=c.get('aggregation_method', 'sum'), update_pocket_coords=False, edge_cutoff=c['edge_cutoff'])
    state = {k[len('ddpm.dynamics.'):]: torch.from_numpy(np.asarray(v)) for k, v in sd.items() if k.startswith('ddpm.dynamics.')}
    dyn.load_state_dict(state)
    dyn = dyn.cuda()
    with pytest.warns(RuntimeWarning, match='half matrix engine'), torch.no_grad():
        eps, _ = dyn(dev(inp['xh_phar']), dev(inp['xh_pocket']), dev(inp['t']), dev(inp['mask_phar']), dev(inp['mask_pocket']))
    got = eps.cpu().numpy()
    assert np.isfinite(got).all() and np.abs(got - want).max() <= EVAL_TOL * max(1.0, float(np.abs(want).max()))


def test_chain_on_an_overflowing_model_equals_the_bf16_engine_chain():
    """A short chain: the guarded call returns exactly what a handle with half_engine = 0 returns (same draws), reports no reset and warns;
    a model inside the range takes the half engine's result and does not warn."""
    from cmdgen_amd.synthetic import make_pockets
    cfg, sd, _ = overflow_case('msg')
    pb = make_pockets(8, 'CA', n_phar=8)
    K = 6
    px, poh = dev(pb.x), dev(pb.one_hot)

    def run_on(h):
        return lambda: h.sample_chain(px, poh, K, noise=None, seed=5, pocket_ids=pb.pocket_index, use_graph=True)
    ref = new_handle(cfg, sd)
    ref.set_option('half_engine', 0)
    ref.set_layout(pb.num_nodes_phar, pb.size)
    ref.set_step_table(K, host_step_table(cfg, K))
    want = ref.sample_chain(px, poh, K, noise=None, seed=5, pocket_ids=pb.pocket_index, use_graph=True)[0].cpu().numpy()
    ref_resets = ref.chain_status()['nan_resets']
    ref.close()
    h = new_handle(cfg, sd)
    h.set_layout(pb.num_nodes_phar, pb.size)
    h.set_step_table(K, host_step_table(cfg, K))
    assert h.half_engine_active()
    raw = run_on(h)()
    assert h.chain_status()['nan_resets'] >= 1, 'the raw half-engine chain is expected to reset on this model'
    del raw
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        (xh, _q, _z), st = h.run_range_guarded(run_on(h), h.chain_status)
    got = xh.cpu().numpy()
    assert st['nan_resets'] == ref_resets and st.get('half_engine_fallback')
    if ref_resets == 0:
        assert any('half matrix engine' in str(x.message) for x in w)
    # (the two handles may pick other tiles - the choice depends on the engine - so sums differ in their last bits: the evaluation tolerance, types exact)
    assert np.abs(got[:, :3] - want[:, :3]).max() <= EVAL_TOL * max(1.0, float(np.abs(want[:, :3]).max())) and np.array_equal(got[:, 3:], want[:, 3:])
    h.close()
    # inside the range: no second run, no warning
    cfg2, sd2, _ = dynamics_case(G2, NAME)
    h2 = new_handle(cfg2, sd2)
    h2.set_layout(pb.num_nodes_phar, pb.size)
    h2.set_step_table(K, host_step_table(cfg2, K))
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        _out, st2 = h2.run_range_guarded(run_on(h2), h2.chain_status)
    assert st2['nan_resets'] == 0 and 'half_engine_fallback' not in st2
    h2.close()
