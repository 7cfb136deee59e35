"""Does a training step depend on the step BEFORE it?  One trainer, two inputs (A, B: other draws and time steps, hence other edge lists) in the
order A A B A B B A ...: every call's gradient against the first call with the same input (deviation relative to each tensor's scale)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import cmdgen_amd  # noqa: F401
from cmdgen_amd import hip_backend
import bench_train as bt
spec = sys.argv[1] if len(sys.argv) > 1 else '-'
if spec != '-':
    hip_backend.DEFAULT_OPTIONS.update(hip_backend.parse_options(spec))
B, first = 64, 7200
dev = torch.device('cuda', 0)
cfg, model, tr = bt.build_trainer(B, 'CA', 'fp32', dev, pipelined=False)
batch = bt.synthetic_batch(B, first, dev)
inp = {}
for name, seed in (('A', 12), ('B', 11), ('C', 13)):
    gen = torch.Generator().manual_seed(seed)
    t_int = torch.randint(1, 501, (B, 1), generator=gen).float().to(dev)
    inp[name] = (t_int, torch.randn((int(batch['num_phar_atoms'].sum()), 11), generator=gen).to(dev))
names = [(n,) + tr.h.param_offset(n) for n, _ in tr.dyn.named_parameters()]
ref = {}
for it, which in enumerate('AABABBACACCBA'):
    tr.loss_and_grad(batch, t_int=inp[which][0], eps=[inp[which][1]])
    g = tr.grad.double().cpu().numpy()
    E, Ec = tr.h.query('train_edges'), tr.h.query('train_coord_edges')
    if which not in ref:
        ref[which] = g
        print('call %2d %s (%d / %d edges): reference' % (it, which, E, Ec)); continue
    r = ref[which]
    rows = sorted(((float(np.abs(g[o:o + c] - r[o:o + c]).max()) / max(float(np.abs(r[o:o + c]).max()), 1e-9), n) for n, o, c in names), reverse=True)
    print('call %2d %s (%d / %d edges): %s' % (it, which, E, Ec, '  '.join('%s %.1e' % (n.replace('egnn.', ''), x) for x, n in rows[:4])), flush=True)
