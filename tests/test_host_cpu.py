"""CPU suite for the host layer: C-ABI symbol check, schedule table, node-count prior, masks,
PDB reader / generate_phars bookkeeping, checkpoint format, failure without a GPU."""
import ctypes
import json
import os
import re
from argparse import Namespace

import numpy as np
import pytest
import torch

from helpers import load_golden
import cmdgen_amd  # noqa: F401
from cmdgen_amd import hip_backend, utils
from cmdgen_amd.constants import dataset_params
from cmdgen_amd.equivariant_diffusion.en_diffusion import DistributionNodes, PredefinedNoiseSchedule
from cmdgen_amd.equivariant_diffusion.dynamics import EGNNDynamics
from cmdgen_amd.equivariant_diffusion.conditional_model import ConditionalDDPM
from cmdgen_amd.lightning_modules import PharPocketDDPM
from cmdgen_amd.synthetic import ModelConfig, make_state_dict, linear_specs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def small_ddpm(H=64, L=2, T=500):
    dyn = EGNNDynamics(phar_nf=8, residue_nf=20, n_dims=3, joint_nf=32, hidden_nf=H, n_layers=L, attention=True,
                       tanh=True, norm_constant=1, inv_sublayers=1, sin_embedding=False, normalization_factor=100,
                       aggregation_method='sum', edge_cutoff=6.0, update_pocket_coords=False)
    return ConditionalDDPM(dynamics=dyn, phar_nf=8, residue_nf=20, n_dims=3, timesteps=T,
                           noise_schedule='polynomial_2', noise_precision=1e-5, loss_type='l2',
                           norm_values=[1, 4], size_histogram=np.ones((30, 70)))


def test_library_exports_every_declared_symbol():
    """The in-tree .so loads (no GPU needed) and exports every function include/cmdgen_hip.h declares."""
    hdr = open(os.path.join(ROOT, 'include', 'cmdgen_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(cmdgen_[a-z_0-9]+)\s*\(', hdr))
    assert len(declared) >= 17
    lib = hip_backend.load_library()
    for name in declared:
        assert hasattr(lib, name), name
    bound = {n for n, _, _ in hip_backend.SYMBOLS}
    assert declared == bound, declared ^ bound
    assert b'gfx950' in lib.cmdgen_version()
    assert ctypes.sizeof(hip_backend.Config) == 21 * 4 and ctypes.sizeof(hip_backend.Counters) == 64


def test_library_source_reads_no_environment_variable():
    """Launch choices are per-handle options (cmdgen_set_option); the C ABI library has no process-global switches."""
    csrc = os.path.join(ROOT, 'cmdgen_amd', 'csrc')
    for f in os.listdir(csrc):
        if f.endswith(('.hip', '.h')):
            assert 'getenv' not in open(os.path.join(csrc, f)).read(), f


def test_product_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    ddpm = small_ddpm()
    with pytest.raises(hip_backend.CmdgenError):
        ddpm.dynamics(torch.zeros(3, 11), torch.zeros(5, 23), torch.zeros(1, 1), torch.zeros(3, dtype=torch.long),
                      torch.zeros(5, dtype=torch.long))
    with pytest.raises(hip_backend.CmdgenError):
        hip_backend.Handle(ModelConfig().as_dict(), 0)
    # and nothing in the product imports the oracle
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'cmdgen_amd')):
        for f in files:
            if f.endswith('.py') and f != 'selftest.py':
                src = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in src.replace('the oracle', ''), f


def test_state_dict_names_match_reference_checkpoint():
    """97 tensors, same names/shapes as the reference's 'ddpm.' sub-tree."""
    cfg = ModelConfig()
    ddpm = small_ddpm(H=256, L=5)
    sd = make_state_dict(cfg, seed=0, prefix='')
    mine = ddpm.state_dict()
    assert set(mine) == set(sd) and len(mine) == 97
    for k, v in sd.items():
        assert tuple(mine[k].shape) == v.shape, k
    ddpm.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    n_params = sum(p.numel() for p in ddpm.parameters())
    assert n_params == 2987815


def test_step_table_is_bit_identical_to_reference_scalars():
    g = load_golden('g1_schedule.npz')
    ddpm = small_ddpm(T=500)
    assert np.array_equal(ddpm.gamma.gamma.numpy(), g['gamma_T500'])
    for K in (5, 50, 500):
        tab = ddpm.step_table(K)
        assert tab.shape == (K + 1, 4)
        assert np.array_equal(tab[:K], g[f'coef_T500_K{K}'])
        assert np.array_equal(tab[K, :3], g['final_T500'])
    sched = PredefinedNoiseSchedule('polynomial_2', 1000, 1e-5)
    t = torch.tensor([[0.0], [0.5004], [1.0]])
    assert np.array_equal(sched(t).numpy().ravel(), g['gamma_T1000'][[0, 500, 1000]])


def test_distribution_nodes_matches_reference():
    g = load_golden('g8_nodes.npz')
    dn = DistributionNodes(g['hist'])
    lp = dn.log_prob_n1_given_n2(torch.from_numpy(g['n1']), torch.from_numpy(g['n2'])).numpy()
    assert np.allclose(lp, g['logp'], rtol=1e-6, atol=1e-7)
    torch.manual_seed(0)
    s = dn.sample_conditional(n1=None, n2=torch.tensor([44, 44, 30, 60]))
    assert s.shape == (4,) and bool(((s >= 3) & (s <= 25)).all())      # support of the histogram
    with pytest.raises(AssertionError):
        dn.sample_conditional(n1=None, n2=None)


def test_mask_helpers():
    m = utils.num_nodes_to_batch_mask(3, torch.tensor([2, 0, 3]), 'cpu')
    assert m.tolist() == [0, 0, 2, 2, 2]
    parts = utils.batch_to_list(torch.arange(5), m)
    assert [p.tolist() for p in parts] == [[0, 1], [2, 3, 4]]
    assert utils.sizes_from_mask(m, 3).tolist() == [2, 0, 3]
    with pytest.raises(ValueError):
        utils.sizes_from_mask(torch.tensor([1, 0]), 2)
    q = utils.Queue(max_len=3)
    for v in (1, 2, 3, 4):
        q.add(v)
    assert len(q) == 3 and q.mean() == 3.0


PDB = """\
ATOM      1  N   ALA A   1      11.104   6.134  -6.504  1.00  0.00           N
ATOM      2  CA  ALA A   1      11.639   6.071  -5.147  1.00  0.00           C
ATOM      3  C   ALA A   1      10.500   6.000  -4.100  1.00  0.00           C
ATOM      4  CA  GLY A   2      14.000   7.500  -3.000  1.00  0.00           C
ATOM      5  CA  TRP A   3      16.500   9.000  -1.000  1.00  0.00           C
ATOM      6  H   TRP A   3      16.900   9.100  -1.100  1.00  0.00           H
HETATM    7  C1  LIG A 101      13.000   7.000  -4.000  1.00  0.00           C
HETATM    8  O   HOH A 201      30.000  30.000  30.000  1.00  0.00           O
END
"""


def test_pdb_reader_and_pocket_selection(tmp_path):
    f = tmp_path / 'p.pdb'
    f.write_text(PDB)
    m = utils.parse_pdb(str(f))
    assert [r.get_resname() for r in m['A'].get_residues()] == ['ALA', 'GLY', 'TRP', 'LIG', 'HOH']
    res = m['A'][(' ', 2, ' ')]
    assert np.allclose(res['CA'].get_coord(), [14.0, 7.5, -3.0])
    assert utils.three_to_one('TRP') == 'W'
    near = utils.get_pocket_from_ligand(m, 'A:101')
    assert [r.get_resname() for r in near] == ['ALA', 'GLY', 'TRP']     # standard amino acids within 8 A


def _hparams(H=64, L=1):
    return dict(outdir='out', dataset='crossdock', datadir='data', batch_size=4, lr=1e-4,
                egnn_params=Namespace(device='cuda', edge_cutoff=6.0, joint_nf=32, hidden_nf=H, n_layers=L,
                                      attention=True, tanh=True, norm_constant=1, inv_sublayers=1,
                                      sin_embedding=False, aggregation_method='sum', normalization_factor=100),
                diffusion_params=Namespace(diffusion_steps=500, diffusion_noise_schedule='polynomial_2',
                                           diffusion_noise_precision=1e-5, diffusion_loss_type='l2',
                                           normalize_factors=[1, 4]),
                num_workers=0, augment_noise=0, augment_rotation=False, clip_grad=True, eval_epochs=50,
                eval_params=Namespace(n_eval_samples=100, eval_batch_size=100), mode='pocket_conditioning',
                node_histogram=np.ones((30, 70)), pocket_representation='CA')


def test_checkpoint_round_trip_lightning_format(tmp_path):
    model = PharPocketDDPM(**_hparams())
    assert all(k.startswith('ddpm.') for k in model.state_dict())
    ck = tmp_path / 'best.ckpt'
    model.save_checkpoint(str(ck))
    again = PharPocketDDPM.load_from_checkpoint(str(ck), map_location='cpu')
    for k, v in model.state_dict().items():
        assert torch.equal(v, again.state_dict()[k])
    assert again.T == 500 and again.phar_nf == 8 and again.aa_nf == 20
    # mode 'joint' builds the base diffusion class over dynamics that also move the pocket (lightning_modules.py:110-139)
    jm = PharPocketDDPM(**{**_hparams(), 'mode': 'joint'})
    assert type(jm.ddpm).__name__ == 'EnVariationalDiffusion' and jm.ddpm.dynamics.update_pocket_coords
    assert set(jm.state_dict()) == set(model.state_dict())
    assert jm.ddpm.get_repaint_schedule(2, 2, 6) == [4, 4, 2]            # en_diffusion.py:649-670 (G9 schedule/r2_j2_T6)


def test_generate_phars_bookkeeping_with_stub_sampler(tmp_path, monkeypatch):
    """Post-parse behaviour of generate_phars (lightning_modules.py:439-541): pocket tensors repeated
    n_samples x, COM restore, and the 'Molecule_k' grouping quirk (Q9) - the sampler itself is stubbed
    here (it needs the GPU; covered by tests/test_hip_parity.py)."""
    f = tmp_path / 'p.pdb'
    f.write_text(PDB)
    model = PharPocketDDPM(**_hparams())
    seen = {}

    def fake_sample(pocket, num_nodes_phar, timesteps=None, **kw):
        seen['pocket'] = {k: v.clone() for k, v in pocket.items()}
        n = len(pocket['size'])
        pm = utils.num_nodes_to_batch_mask(n, num_nodes_phar, 'cpu')
        x = torch.arange(len(pm) * 3, dtype=torch.float32).view(-1, 3)
        types = torch.tensor([0, 4, 4, 1, 4, 0][:len(pm)])
        xh = torch.cat([x, torch.nn.functional.one_hot(types, 8).float()], 1)
        shift = torch.tensor([1.0, -2.0, 3.0])
        xp = torch.cat([pocket['x'] + shift, pocket['one_hot'].float()], 1)     # sampler translated the pocket
        xh[:, :3] += shift
        return xh, xp, pm, pocket['mask']
    monkeypatch.setattr(model.ddpm, 'sample_given_pocket', fake_sample)
    out = model.generate_phars(str(f), 2, pocket_ids=['A:1', 'A:2', 'A:3'], num_nodes_phar=torch.tensor([3, 3]))
    p = seen['pocket']
    assert p['x'].shape == (6, 3) and p['size'].tolist() == [3, 3] and p['mask'].tolist() == [0, 0, 0, 1, 1, 1]
    assert p['one_hot'].argmax(1).tolist() == [dataset_params['crossdock']['aa_encoder'][a] for a in 'AGW'] * 2
    assert sorted(out) == ['Molecule_1', 'Molecule_2', 'Molecule_3']           # k-th point of ALL samples
    assert sorted(out['Molecule_1']) == ['Aromatic', 'Hydrophobe']
    assert len(out['Molecule_2']['Acceptor']) == 2
    # the sampler's translation is undone: first point of sample 0 is back at (0,1,2)
    assert torch.allclose(out['Molecule_1']['Aromatic'][0], torch.tensor([0.0, 1.0, 2.0]), atol=1e-5)
    json.dumps({m: {t: [c.tolist() for c in cs] for t, cs in d.items()} for m, d in out.items()})


def test_generate_phars_cli_flags():
    from cmdgen_amd.generate_phars import build_parser
    a = build_parser().parse_args(['ckpt', '--pdbfile', 'x.pdb', '--resi_list', 'A:1', 'A:2', '--timesteps', '50'])
    assert a.n_samples == 20 and a.num_nodes_phar == 3 and a.resamplings == 10 and a.jump_length == 1
    assert a.resi_list == ['A:1', 'A:2'] and a.timesteps == 50 and not a.sanitize and not a.relax


def test_linear_specs_cover_all_weights():
    """2 987 314 trainable parameters + the frozen 501-entry gamma table."""
    n = sum(o * i + (o if b else 0) for o, i, b in linear_specs(ModelConfig()).values())
    assert n == 2987314 and n + 501 == 2987815


def test_dataset_npz_schema_centering_and_collate(tmp_path):
    """dataset.py:7-64: split by mask, per-complex joint centring, collate to a flat batch with float masks (Q13)."""
    from cmdgen_amd.dataset import ProcessedLigandPharPocketDataset, write_synthetic_npz
    f = tmp_path / 'val.npz'
    pb = write_synthetic_npz(str(f), n_complexes=5, seed=2)
    ds = ProcessedLigandPharPocketDataset(str(f))
    assert len(ds) == 5 and ds.data['num_pocket_nodes'].tolist() == pb.size.tolist()
    for i in range(5):
        it = ds[i]
        allx = torch.cat([it['phar_coords'], it['pocket_c_alpha']])
        assert float(allx.mean(0).abs().max()) < 1e-4                      # centred on the joint COG
        assert it['phar_one_hot'].shape[1] == 8 and it['pocket_one_hot'].shape[1] == 20
    batch = ds.collate_fn([ds[3], ds[1]])
    assert batch['names'] == ['complex_3', 'complex_1']
    assert batch['phar_mask'].dtype == torch.float32 and batch['phar_mask'].unique().tolist() == [0.0, 1.0]
    assert batch['num_pocket_nodes'].tolist() == [int(pb.size[3]), int(pb.size[1])]
    assert len(batch['pocket_c_alpha']) == int(pb.size[3] + pb.size[1])
    raw = ProcessedLigandPharPocketDataset(str(f), center=False)
    assert float(torch.cat([raw[0]['phar_coords'], raw[0]['pocket_c_alpha']]).mean(0).abs().max()) > 1.0


def test_consensus_posp_matches_reference_script(tmp_path):
    """get_phar/GMM_json.py (run unmodified by tests/golden/make_golden_posp.py) vs cmdgen_amd.get_phar: same
    GMM (scikit-learn, random_state=42), same typing rule, same .posp text; the lines parse the way
    GCPG/utils/file_utils.py:67-102 splits them."""
    import json
    from cmdgen_amd import get_phar
    cases = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'g10_posp.json')))['cases']
    assert len(cases) == 3
    for c in cases:
        clusters = get_phar.gmm_consensus(c['input'])
        text = ''.join(line + '\n' for line in get_phar.posp_lines(clusters))
        assert text == c['posp']
        for line in text.strip().split('\n'):
            types, x, y, z = line.strip().split(' ')
            assert types in get_phar.IDX2PHAR.values() and all(np.isfinite(float(v)) for v in (x, y, z))
    assert any(len(c['posp'].splitlines()) < 7 for c in cases)          # a 'NegIonizable' cluster is dropped (quirk kept)
    src = tmp_path / 'in.json'
    src.write_text(json.dumps(cases[0]['input']))
    lines = get_phar.main([str(src), '--out', str(tmp_path / 'o.posp')])
    assert (tmp_path / 'o.posp').read_text() == cases[0]['posp'] and len(lines) == len(cases[0]['posp'].splitlines())


def test_per_sample_table_matches_the_tensor_op_scalars():
    """training.per_sample_table (numpy on the host: the `tab` argument of cmdgen_train_noise / cmdgen_train_loss) against the
    same per-sample scalars formed with ConditionalDDPM's own tensor operations, t = 0 and t = T included."""
    from cmdgen_amd.training import per_sample_table
    model = PharPocketDDPM(**_hparams())
    ddpm = model.ddpm
    B = 7
    t_int = torch.tensor([0., 1., 2., 17., 250., 499., 500.][:B]).reshape(B, 1)
    n_phar, n_pocket = torch.tensor([3, 5, 8, 12, 15, 20, 9]), torch.tensor([30, 44, 51, 38, 60, 47, 33])
    tab = per_sample_table(ddpm.gamma.gamma.detach().numpy(), ddpm.size_distribution._table(1, torch.device('cpu')).numpy(), ddpm.T,
                           ddpm.n_dims, ddpm.norm_values, t_int, n_phar.numpy(), n_pocket.numpy())
    assert tuple(tab.shape) == (12, B) and tab.dtype == torch.float32
    x = torch.zeros(B, 3)
    s, t = (t_int - 1) / ddpm.T, t_int / ddpm.T
    gamma_s, gamma_t = ddpm.inflate_batch_array(ddpm.gamma(s), x), ddpm.inflate_batch_array(ddpm.gamma(t), x)
    ones = torch.ones((B, 1))
    want = [ddpm.alpha(gamma_t, x).reshape(-1), ddpm.sigma(gamma_t, x).reshape(-1), (t_int == 0).float().reshape(-1),
            (1 - ddpm.SNR(gamma_s - gamma_t)).reshape(-1), ddpm.alpha(ddpm.gamma(ones), x).reshape(-1), ddpm.sigma(ddpm.gamma(ones), x).reshape(-1),
            -ddpm.log_constants_p_x_given_z0(n_nodes=n_phar, device='cpu'), ddpm.delta_log_px(n_phar).float(), ddpm.log_pN(n_phar, n_pocket),
            t_int.reshape(-1), t.reshape(-1), ddpm.sigma(gamma_t, x).reshape(-1) * ddpm.norm_values[1]]
    for i, w in enumerate(want):
        w = torch.as_tensor(w, dtype=torch.float32).reshape(-1)
        assert torch.allclose(tab[i], w, rtol=2e-6, atol=1e-7), (i, tab[i], w)


def test_gamma_network_host_copy_leaves_the_global_rng_alone():
    """GammaNetwork.cpu_copy() constructs a fresh network (PositiveLinear.__init__ draws from the global generator): it must not advance the
    caller's stream - the reference's sampler does not touch it, so host-side randn after a sampling call stays reproducible against it."""
    from cmdgen_amd.equivariant_diffusion.en_diffusion import GammaNetwork
    g = GammaNetwork()
    torch.manual_seed(123)
    want = torch.rand(4)
    torch.manual_seed(123)
    g.cpu_copy(); g.table(50)
    assert torch.equal(torch.rand(4), want)


def test_joint_sampler_refuses_a_learned_schedule_off_the_T_grid():
    """mode 'joint' + noise_schedule='learned': the joint chain's op table reads gamma from the network's tabulation on the T-grid, which
    equals the reference's gamma(step / K) only when K divides T - any other K is refused before the handle is touched (no GPU needed)."""
    from cmdgen_amd.equivariant_diffusion.en_diffusion import EnVariationalDiffusion
    dyn = EGNNDynamics(phar_nf=8, residue_nf=20, n_dims=3, joint_nf=16, hidden_nf=64, n_layers=1, attention=True, tanh=True, norm_constant=1,
                       inv_sublayers=1, normalization_factor=100, aggregation_method='sum', edge_cutoff=6.0, update_pocket_coords=True)
    ddpm = EnVariationalDiffusion(dynamics=dyn, phar_nf=8, residue_nf=20, n_dims=3, timesteps=500, noise_schedule='learned', noise_precision=1e-5,
                                  loss_type='vlb', norm_values=[1, 4], size_histogram=np.ones((30, 70)))
    with pytest.raises(NotImplementedError, match='divide T'):
        ddpm._joint_handle(np.array([8]), np.array([40]), timesteps=7)
