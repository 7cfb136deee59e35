#!/usr/bin/env python3
"""Throughput of the training path's generic fp32 MFMA GEMM (k_sgemm) on the shapes the training step uses."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _opts  # noqa: E401,F401  (CMDGEN_OPTIONS -> handle options)
import cmdgen_amd  # noqa: E402,F401
from cmdgen_amd import hip_backend  # noqa: E402
from cmdgen_amd.synthetic import ModelConfig  # noqa: E402

h = hip_backend.Handle(ModelConfig(hidden_nf=64, n_layers=1).as_dict(), 0)
E, H = 25600, 256
shapes = [('fwd  y=xW^T   [E,256]x[256,256]^T', False, True, (E, H), (H, H), 1),
          ('dgrad dx=dyW  [E,256]x[256,256]', False, False, (E, H), (H, H), 1),
          ('wgrad dW=dy^Tx [256,E]x[E,256]', True, False, (E, H), (E, H), 0),
          ('node fwd      [3776,256]x[256,256]^T', False, True, (3776, H), (H, H), 1),
          ('node wgrad    [256,3776]x[3776,256]', True, False, (3776, H), (3776, H), 0),
          ('fwd 4E        [102400,256]x[256,256]^T', False, True, (4 * E, H), (H, H), 1)]
for name, ta, tb, sa, sb, split in shapes:
    A = torch.randn(*sa, device='cuda'); B = torch.randn(*sb, device='cuda')
    M = sa[1] if ta else sa[0]; K = sa[0] if ta else sa[1]; N = sb[0] if tb else sb[1]
    C = torch.zeros(M, N, device='cuda')
    for _ in range(3):
        h.debug_sgemm(A, B, ta=ta, tb=tb, C_out=C, accumulate=True, split_k=split)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); reps = 50
    for _ in range(reps):
        h.debug_sgemm(A, B, ta=ta, tb=tb, C_out=C, accumulate=True, split_k=split)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f'{name:45s} {dt * 1e6:8.1f} us  {2.0 * M * N * K / dt / 1e12:6.1f} TF/s')
