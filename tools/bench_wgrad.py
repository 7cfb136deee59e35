#!/usr/bin/env python3
"""Weight-gradient launch dW += dY^T X on the training step's shapes: fp32 instruction (mode 0), bf16 operands (1), three bf16 pieces (3).
Single products through cmdgen_debug_wgrad (the step groups the seven node-level products of a block into one launch)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _opts  # noqa: E401,F401  (CMDGEN_OPTIONS -> handle options)
from cmdgen_amd import hip_backend  # noqa: E402
from cmdgen_amd.synthetic import ModelConfig  # noqa: E402

h = hip_backend.Handle(ModelConfig(hidden_nf=64, n_layers=1).as_dict(), 0)
g = torch.Generator().manual_seed(0)
for K, M, N in [(3721, 256, 256), (3721, 256, 512), (16127, 256, 256), (36147, 256, 256), (156316, 256, 256)]:
    dY, X = torch.randn(K, M, generator=g).cuda(), torch.randn(K, N, generator=g).cuda()
    want = dY.double().t() @ X.double()
    line = f'K={K:6d} M={M} N={N}:'
    for mode in (0, 1, 3):
        dW, db = torch.zeros(M, N, device='cuda'), torch.zeros(M, device='cuda')
        h.debug_wgrad(dY, X, dW, db, mode=mode)
        torch.cuda.synchronize()
        err = float((dW.double() - want).abs().max() / want.abs().max())
        t0 = time.perf_counter(); reps = 30
        for _ in range(reps): h.debug_wgrad(dY, X, dW, db, mode=mode)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        line += f'  mode {mode}: {dt * 1e6:7.1f} us {2.0 * M * N * K / dt / 1e12:6.1f} TF/s err {err:.1e}'
    print(line, flush=True)
