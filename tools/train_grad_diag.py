"""Which part of the training step limits its gradient accuracy?  The step at the benchmark's size (64 complexes) against autograd through the
oracle in FLOAT64, for a list of option sets (CMDGEN option strings, ';'-separated; 'fp32' = cmdgen_set_gemm_mode(0)):
    python tools/train_grad_diag.py "-" "train_half=0" "wgrad_split=0" "fp32"
Prints, per set, the tensors with the largest max|dg| / max|g| (the oracle's own fp32-vs-fp64 difference is ~5e-7 here)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import cmdgen_amd  # noqa: F401
from cmdgen_amd import hip_backend
from cmdgen_amd.synthetic import make_state_dict, make_training_batch
from oracle import ref_cpu          # (diagnostic tool: the oracle is the checker here, as in tests/)
import bench_train as bt

B, first = 64, 7200
dev = torch.device('cuda', 0)
gen = torch.Generator().manual_seed(11)
t_int = torch.randint(1, 501, (B, 1), generator=gen).float()
nb = make_training_batch(B, first, 'CA')
nl_tot = int(nb['num_phar_atoms'].sum())
eps0 = torch.randn((nl_tot, 11), generator=gen)
want = None
for spec in (sys.argv[1:] or ['-']):
    hip_backend.DEFAULT_OPTIONS.clear()
    if spec not in ('-', 'fp32'):
        hip_backend.DEFAULT_OPTIONS.update(hip_backend.parse_options(spec))
    cfg, model, tr = bt.build_trainer(B, 'CA', 'fp32', dev, pipelined=False)
    if spec == 'fp32':
        tr.h.set_gemm_mode(False)
    batch = bt.synthetic_batch(B, first, dev)
    loss, nll, info = tr.loss_and_grad(batch, t_int=t_int.to(dev), eps=[eps0.to(dev)])
    grad = tr.grad.double().cpu().numpy()
    if want is None:
        dt = torch.float64
        ref_cpu.FLOAT = dt
        sd = make_state_dict(cfg, seed=0)
        t = lambda v: torch.from_numpy(np.ascontiguousarray(v))
        phar = {'x': t(nb['phar_coords']).to(dt), 'one_hot': t(nb['phar_one_hot']).to(dt), 'size': t(nb['num_phar_atoms']), 'mask': t(nb['phar_mask'])}
        pocket = {'x': t(nb['pocket_c_alpha']).to(dt), 'one_hot': t(nb['pocket_one_hot']).to(dt), 'size': t(nb['num_pocket_nodes']), 'mask': t(nb['pocket_mask'])}
        p = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in ref_cpu.to_torch_params(sd).items()}
        leaves = {k: v.clone().requires_grad_(True) for k, v in p.items() if k.startswith('dynamics.')}
        p2 = dict(p); p2.update(leaves)
        terms = ref_cpu.ddpm_forward(p2, cfg.as_dict(), phar, pocket, t_int.to(dt), [eps0.to(dt)], training=True, histogram=np.ones((30, 500)))
        w = ref_cpu.nll_from_terms(terms, cfg.as_dict(), phar['size'], pocket['size'], training=True)
        w.mean(0).backward()
        want = {k: (v.grad.numpy().reshape(-1) if v.grad is not None else None) for k, v in leaves.items()}
        ref_cpu.FLOAT = torch.float32
        print('oracle (float64) loss %.9f' % float(w.mean()))
    rows = []
    for name, g in want.items():
        off, cnt = tr.h.param_offset(name[len('dynamics.'):])
        gw = np.zeros(cnt) if g is None else g
        rows.append((float(np.abs(grad[off:off + cnt] - gw).max()) / max(float(np.abs(gw).max()), 1e-6), name[len('dynamics.egnn.'):] if 'egnn.' in name else name))
    rows.sort(reverse=True)
    print('[%s] loss %.9f | worst: %s' % (spec, float(loss), '  '.join('%s %.1e' % (n, r) for r, n in rows[:6])), flush=True)
    del tr, model
