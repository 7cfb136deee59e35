"""cmdgen_set_option / cmdgen_get_option (include/cmdgen_hip.h): per-handle launch choices; the library reads no environment variable."""
import os

import numpy as np
import pytest
import torch

from helpers import load_golden, cases_of, dynamics_case
from cmdgen_amd import hip_backend
from test_hip_parity_r2 import dev, new_handle, EVAL_TOL

pytestmark = pytest.mark.gpu
G2 = load_golden('g2_dynamics.npz')


def test_options_are_per_handle_and_reversible(monkeypatch):
    name = [n for n in cases_of(G2) if '_h256_' in n][-1]
    cfg, sd, inp = dynamics_case(G2, name)
    want = G2[name + '/eps_phar']
    monkeypatch.setenv('CMDGEN_EDGE_MT', '64')            # a relic of earlier rounds: must have no effect any more
    a, b = new_handle(cfg, sd), new_handle(cfg, sd)
    for h in (a, b):
        h.set_layout(G2[name + '/num_nodes_phar'], G2[name + '/pocket_size'])
    own = a.query('edge_mt')
    assert own == b.query('edge_mt') and a.get_option('edge_mt') is None
    a.set_option('edge_mt', 64); a.set_option('node_mt', 32); a.set_option('dead_skip', 0)
    assert a.query('edge_mt') == 64 and a.query('node_mt') == 32 and a.query('dead_skip') == 0 and a.get_option('edge_mt') == 64
    assert b.query('edge_mt') == own and b.query('dead_skip') == 2                      # the other handle is untouched
    for h in (a, b):
        eps, _ = h.dynamics_forward(dev(inp['xh_phar']), dev(inp['xh_pocket']), dev(inp['t']))
        assert float(np.abs(eps.cpu().numpy() - want).max()) <= EVAL_TOL * max(1.0, float(np.abs(want).max()))
    a.set_option('edge_mt', None); a.set_option('node_mt', None)
    assert a.query('edge_mt') == own and a.get_option('edge_mt') is None
    with pytest.raises(hip_backend.CmdgenError, match='unknown option'):
        a.set_option('edge_tiles', 1)
    a.close(); b.close()


def test_eight_wave_node_tile_against_the_four_wave_one():
    """kernels_node16w.hip (16-row node tiles of H = 256 on eight waves) against k_node<256, 16> on the same split engine: the same MFMAs in the
    same order on every accumulator - the two agree as closely as two runs of either do (the segment sums upstream add a receiver's tile
    partials with float atomics, so an evaluation is reproducible to the last bits, not bit for bit) - and both sit inside the evaluation
    tolerance of the reference's result."""
    for name in [n for n in cases_of(G2) if '_h256_' in n]:
        cfg, sd, inp = dynamics_case(G2, name)
        want = G2[name + '/eps_phar']
        got = []
        for on in (1, 0, 0):
            h = new_handle(cfg, sd)
            h.set_option('node_mt', 16); h.set_option('node64', 0); h.set_option('node16w', on)
            h.set_layout(G2[name + '/num_nodes_phar'], G2[name + '/pocket_size'])
            assert h.query('node_mt') == 16 and h.query('node16w') == on
            eps, _ = h.dynamics_forward(dev(inp['xh_phar']), dev(inp['xh_pocket']), dev(inp['t']))
            got.append(eps.cpu().numpy())
            h.close()
        scale = max(1.0, float(np.abs(want).max()))
        run_to_run = float(np.abs(got[1] - got[2]).max())
        assert float(np.abs(got[0] - got[1]).max()) <= max(4.0 * run_to_run, 2e-6 * scale), name
        for g in got:
            assert float(np.abs(g - want).max()) <= EVAL_TOL * scale
