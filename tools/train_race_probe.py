"""Is the training step's gradient reproducible call to call?  One trainer, the same batch / draws / time steps, N repetitions of loss_and_grad:
per repetition the largest deviation of any tensor from the FIRST call's gradient, relative to that tensor's scale (float atomics reorder sums:
~1e-6 is expected; 1e-4 and more means two kernels raced).   python tools/train_race_probe.py [N] [option string]   (DIAG_STREAM=1: a torch side stream)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import cmdgen_amd  # noqa: F401
from cmdgen_amd import hip_backend
import bench_train as bt
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
spec = sys.argv[2] if len(sys.argv) > 2 else '-'
if spec != '-':
    hip_backend.DEFAULT_OPTIONS.update(hip_backend.parse_options(spec))
B, first = 64, 7200
dev = torch.device('cuda', 0)
cfg, model, tr = bt.build_trainer(B, 'CA', 'fp32', dev, pipelined=False)
batch = bt.synthetic_batch(B, first, dev)
gen = torch.Generator().manual_seed(12)
t_int = torch.randint(1, 501, (B, 1), generator=gen).float().to(dev)
eps0 = torch.randn((int(batch['num_phar_atoms'].sum()), 11), generator=gen).to(dev)
names = [(n,) + tr.h.param_offset(n) for n, _ in tr.dyn.named_parameters()]
st = torch.cuda.Stream() if os.environ.get('DIAG_STREAM', '0') == '1' else torch.cuda.current_stream()
ref = None
bad = 0
with torch.cuda.stream(st):
    for it in range(N):
        tr.loss_and_grad(batch, t_int=t_int, eps=[eps0])
        if os.environ.get('DIAG_SLEEP'):
            torch.cuda.synchronize(); import time; time.sleep(float(os.environ['DIAG_SLEEP']))
        g = tr.grad.double().cpu().numpy()
        if ref is None:
            ref = g; continue
        rows = sorted(((float(np.abs(g[o:o + c] - ref[o:o + c]).max()) / max(float(np.abs(ref[o:o + c]).max()), 1e-9), n) for n, o, c in names), reverse=True)
        flag = rows[0][0] > 2e-5
        bad += flag
        if flag or it < 3:
            print('rep %2d: %s' % (it, '  '.join('%s %.1e' % (n.replace('egnn.', ''), r) for r, n in rows[:4])), flush=True)
print('[%s] %d of %d repetitions deviate by more than 2e-5 from the first call' % (spec, bad, N - 1))
