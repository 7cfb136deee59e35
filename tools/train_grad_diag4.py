"""Diagnostic: a second trainer created while the first is alive - its first and second call against the first trainer's gradient (HIP vs HIP)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import cmdgen_amd  # noqa: F401
import bench_train as bt
from cmdgen_amd import hip_backend
if len(sys.argv) > 1 and sys.argv[1] != '-':
    hip_backend.DEFAULT_OPTIONS.update(hip_backend.parse_options(sys.argv[1]))
MAIN = torch.cuda.Stream() if os.environ.get('DIAG_STREAM', '0') == '1' else None
if MAIN is not None:
    torch.cuda.set_stream(MAIN)
B, first = 64, 7200
dev = torch.device('cuda', 0)
cfg, model, tr = bt.build_trainer(B, 'CA', 'fp32', dev, pipelined=False)
batch = bt.synthetic_batch(B, first, dev)
names = [(n,) + tr.h.param_offset(n) for n, _ in tr.dyn.named_parameters()]
def cmp(g, r):
    rows = sorted(((float(np.abs(g[o:o + c] - r[o:o + c]).max()) / max(float(np.abs(r[o:o + c]).max()), 1e-9), n) for n, o, c in names), reverse=True)
    return '  '.join('%s %.1e' % (n.replace('egnn.', ''), x) for x, n in rows[:3])
SEEDS = tuple(int(x) for x in os.environ.get('DIAG_SEEDS', '11,14,15').split(','))
EXTRA = [torch.cuda.Stream() for _ in range(int(os.environ.get('DIAG_EXTRA_STREAMS', '0')))]
for seed in SEEDS:
    gen = torch.Generator().manual_seed(seed)
    t_int = torch.randint(1, 501, (B, 1), generator=gen).float().to(dev)
    eps0 = torch.randn((int(batch['num_phar_atoms'].sum()), 11), generator=gen).to(dev)
    l0 = tr.loss_and_grad(batch, t_int=t_int, eps=[eps0])[0]
    g0 = tr.grad.double().cpu().numpy()
    cfg2, model2, tr2 = bt.build_trainer(B, 'CA', 'fp32', dev, pipelined=False)
    same_w = bool(torch.equal(tr2.theta, tr.theta))
    l1 = tr2.loss_and_grad(batch, t_int=t_int, eps=[eps0])[0]
    g1 = tr2.grad.double().cpu().numpy()
    l2 = tr2.loss_and_grad(batch, t_int=t_int, eps=[eps0])[0]
    g2 = tr2.grad.double().cpu().numpy()
    print('seed %d: theta equal %s; loss %.9f / %.9f / %.9f; E %d vs %d\n   2nd trainer 1st call vs 1st trainer: %s\n   2nd trainer 2nd call vs 1st trainer: %s' % (
        seed, same_w, float(l0), float(l1), float(l2), tr.h.query('train_edges'), tr2.h.query('train_edges'), cmp(g1, g0), cmp(g2, g0)), flush=True)
    del tr2, model2
