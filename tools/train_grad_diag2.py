"""Diagnostic: the bench-size oracle comparison of tests/test_hip_train.py, seed by seed, on ONE trainer (repeated calls) and on fresh trainers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import cmdgen_amd  # noqa: F401
from cmdgen_amd.synthetic import make_state_dict, make_training_batch, min_cutoff_margin
from oracle import ref_cpu
import bench_train as bt
B, first = 64, 7200
dev = torch.device('cuda', 0)
nb = make_training_batch(B, first, 'CA')
nl_tot = int(nb['num_phar_atoms'].sum())
t = lambda v: torch.from_numpy(np.ascontiguousarray(v))
def oracle(cfg, t_int, eps0, dt=torch.float32):
    ref_cpu.FLOAT = dt
    sd = make_state_dict(cfg, seed=0)
    phar = {'x': t(nb['phar_coords']).to(dt), 'one_hot': t(nb['phar_one_hot']).to(dt), 'size': t(nb['num_phar_atoms']), 'mask': t(nb['phar_mask'])}
    pocket = {'x': t(nb['pocket_c_alpha']).to(dt), 'one_hot': t(nb['pocket_one_hot']).to(dt), 'size': t(nb['num_pocket_nodes']), 'mask': t(nb['pocket_mask'])}
    p = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in ref_cpu.to_torch_params(sd).items()}
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items() if k.startswith('dynamics.')}
    p2 = dict(p); p2.update(leaves)
    terms = ref_cpu.ddpm_forward(p2, cfg.as_dict(), phar, pocket, t_int.to(dt), [eps0.to(dt)], training=True, histogram=np.ones((30, 500)))
    w = ref_cpu.nll_from_terms(terms, cfg.as_dict(), phar['size'], pocket['size'], training=True)
    w.mean(0).backward()
    ref_cpu.FLOAT = torch.float32
    return {k: (v.grad.double().numpy().reshape(-1) if v.grad is not None else None) for k, v in leaves.items()}
def worst(tr, want):
    grad = tr.grad.double().cpu().numpy()
    rows = []
    for name, g in want.items():
        off, cnt = tr.h.param_offset(name[len('dynamics.'):])
        gw = np.zeros(cnt) if g is None else g
        rows.append((float(np.abs(grad[off:off + cnt] - gw).max()) / max(float(np.abs(gw).max()), 1e-6), name.replace('dynamics.egnn.', '')))
    rows.sort(reverse=True)
    return '  '.join('%s %.1e' % (n, r) for r, n in rows[:4])
if __name__ == '__main__':
    cfg, model, tr = bt.build_trainer(B, 'CA', 'fp32', dev, pipelined=False)
    batch = bt.synthetic_batch(B, first, dev)
    for seed in (11, 12, 13, 14):
        gen = torch.Generator().manual_seed(seed)
        t_int = torch.randint(1, 501, (B, 1), generator=gen).float()
        eps0 = torch.randn((nl_tot, 11), generator=gen)
        want = oracle(cfg, t_int, eps0)
        tr.loss_and_grad(batch, t_int=t_int.to(dev), eps=[eps0.to(dev)])
        z = tr._last_fused['z_t'][:, :3].cpu().numpy(); q = tr._last_fused['xh_pocket'][:, :3].cpu().numpy()
        margin = min_cutoff_margin(np.concatenate([z, q]), np.concatenate([nb['phar_mask'], nb['pocket_mask']]), 6.0)
        print('seed %d margin %.1e | same trainer : %s' % (seed, margin, worst(tr, want)), flush=True)
        cfg2, model2, tr2 = bt.build_trainer(B, 'CA', 'fp32', dev, pipelined=False)
        tr2.loss_and_grad(batch, t_int=t_int.to(dev), eps=[eps0.to(dev)])
        print('                         | fresh trainer: %s' % worst(tr2, want), flush=True)
        want64 = oracle(cfg, t_int, eps0, torch.float64)
        print('                         | fresh vs f64 : %s' % worst(tr2, want64), flush=True)
        del tr2, model2
