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
