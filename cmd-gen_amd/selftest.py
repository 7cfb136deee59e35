"""smoke(): one small invocation of the hot path on cuda:0, checked against the oracle."""
import os
import sys

import numpy as np
import torch


def smoke_check(verbose=True):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import ref_cpu          # checker only
    from .synthetic import ModelConfig, make_state_dict, make_pockets
    from . import hip_backend
    assert torch.cuda.is_available(), 'smoke() needs an MI355X'
    cfg = ModelConfig()                 # the shipped C-alpha model: H=256, L=5
    sd = make_state_dict(cfg, seed=0)
    pb = make_pockets(4, 'CA', n_phar=8)
    K = 4
    dev = torch.device('cuda:0')
    h = hip_backend.Handle(cfg.as_dict(), 0)
    h.load_state_dict(sd)
    h.set_layout(pb.num_nodes_phar, pb.size)
    nl = int(pb.num_nodes_phar.sum())
    g = torch.Generator().manual_seed(0)
    noise = torch.randn((K + 2, nl, 3 + cfg.phar_nf), generator=g)
    xh_phar, xh_pocket, _ = h.sample_chain(torch.from_numpy(pb.x).to(dev), torch.from_numpy(pb.one_hot).to(dev),
                                           K, noise=noise.to(dev), use_graph=True)
    st = h.chain_status()
    tape = iter(noise)
    p = ref_cpu.to_torch_params(sd)
    pocket = {'x': torch.from_numpy(pb.x), 'one_hot': torch.from_numpy(pb.one_hot),
              'size': torch.from_numpy(pb.size), 'mask': torch.from_numpy(pb.mask)}
    with torch.no_grad():
        ref_phar, ref_pocket, _, _ = ref_cpu.sample_given_pocket(p, cfg.as_dict(), pocket, pb.num_nodes_phar,
                                                                timesteps=K, noise=lambda shape: next(tape))
    got = xh_phar.cpu().numpy()
    want = ref_phar.numpy()
    rms = float(np.sqrt(np.mean((got[:, :3] - want[:, :3]) ** 2)))
    scale = max(1.0, float(np.abs(want[:, :3]).max()))
    if verbose:
        print(f'smoke: coords RMS vs oracle {rms:.3e} (scale {scale:.1f}), types equal: '
              f'{bool(np.array_equal(got[:, 3:], want[:, 3:]))}, status {st}')
    assert rms < 1e-4 * scale, rms
    assert np.array_equal(got[:, 3:], want[:, 3:])
    assert st['max_rel_com_error'] < 1e-2
    return rms
