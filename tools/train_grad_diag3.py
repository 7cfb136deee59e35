"""Diagnostic: first loss_and_grad of FRESH trainers against the float64 oracle, several times per option set (is the deviation a race?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import cmdgen_amd  # noqa: F401
from cmdgen_amd import hip_backend
from cmdgen_amd.synthetic import make_state_dict, make_training_batch
from oracle import ref_cpu
import bench_train as bt
from train_grad_diag2 import oracle, worst, nb, nl_tot, B, first, dev   # (runs that script's loop once on import: harmless)

use_stream = os.environ.get('DIAG_STREAM', '0') == '1'
for spec in sys.argv[1:] or ['-']:
    hip_backend.DEFAULT_OPTIONS.clear()
    if spec != '-':
        hip_backend.DEFAULT_OPTIONS.update(hip_backend.parse_options(spec))
    for seed in (13, 14):
        gen = torch.Generator().manual_seed(seed)
        t_int = torch.randint(1, 501, (B, 1), generator=gen).float()
        eps0 = torch.randn((nl_tot, 11), generator=gen)
        cfg0 = bt.build_trainer(B, 'CA', 'fp32', dev, pipelined=False)[0]
        want = oracle(cfg0, t_int, eps0, torch.float64)
        for rep in range(4):
            cfg, model, tr = bt.build_trainer(B, 'CA', 'fp32', dev, pipelined=False)
            batch = bt.synthetic_batch(B, first, dev)
            st = torch.cuda.Stream() if use_stream else torch.cuda.current_stream()
            with torch.cuda.stream(st):
                tr.loss_and_grad(batch, t_int=t_int.to(dev), eps=[eps0.to(dev)])
                torch.cuda.synchronize()
                first_call = worst(tr, want)
                tr.loss_and_grad(batch, t_int=t_int.to(dev), eps=[eps0.to(dev)])
                torch.cuda.synchronize()
                second_call = worst(tr, want)
            print('[%s] seed %d rep %d stream=%s | 1st: %s\n%s| 2nd: %s' % (spec, seed, rep, use_stream, first_call, ' ' * 30, second_call), flush=True)
            del tr, model
