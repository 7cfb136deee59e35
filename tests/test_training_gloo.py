"""world_size-2 gloo test of data-parallel TRAINING (CPU): every rank takes the gradient of the training loss on
its own half of a batch, HipTrainer's staged backward + chunked all-reduce (and the single-bucket path) average the
flat gradient, and the result equals the
gradient of the loss on the whole batch (loss = mean over complexes, lightning_modules.py:254; DDP averaging,
train.py:111-121).  The per-rank gradient comes from autograd through the oracle here (test infrastructure standing
in for cmdgen_train_backward, whose own parity is tests/test_hip_train.py on the GPU)."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import cmdgen_amd  # noqa: E402,F401
from cmdgen_amd.synthetic import linear_specs  # noqa: E402
from cmdgen_amd.training import HipTrainer  # noqa: E402


def flat_grad_of_subbatch(sel):
    """oracle autograd gradient of the l2 training loss on the complexes `sel` of the G6 batch, flattened in the
    state_dict order (the layout cmdgen_param_offset describes)."""
    from helpers import load_golden, loss_case
    from oracle import ref_cpu
    g6 = load_golden('g6_loss.npz')
    cfg, sd, phar, pocket, hist = loss_case(g6)
    B = len(phar['size'])
    keep = torch.zeros(B, dtype=torch.bool); keep[sel] = True
    remap = torch.cumsum(keep.long(), 0) - 1

    def sub(d):
        m = keep[d['mask']]
        return {'x': d['x'][m], 'one_hot': d['one_hot'][m], 'size': d['size'][keep], 'mask': remap[d['mask'][m]]}
    ph, pk = sub(phar), sub(pocket)
    p = ref_cpu.to_torch_params(sd)
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items() if k.startswith('dynamics.')}
    p2 = dict(p); p2.update(leaves)
    eps = torch.from_numpy(g6['eps0'])[keep[phar['mask']]]
    terms = ref_cpu.ddpm_forward(p2, cfg.as_dict(), ph, pk, torch.from_numpy(g6['t_int'])[keep], [eps], training=True,
                                 histogram=hist)
    nll = ref_cpu.nll_from_terms(terms, cfg.as_dict(), ph['size'], pk['size'], training=True)
    nll.mean(0).backward()
    parts = []
    for name, (fo, fi, has_bias) in linear_specs(cfg).items():
        for suffix, shape in (('.weight', (fo, fi)),) + ((('.bias', (fo,)),) if has_bias else ()):
            gr = leaves['dynamics.' + name + suffix].grad
            parts.append((torch.zeros(shape) if gr is None else gr).reshape(-1))
    return torch.cat(parts)


class StagedStub(HipTrainer):
    """HipTrainer's data-parallel logic (grad_chunks / _backward / _allreduce, the product code under test) on top of a
    fake handle whose staged backward pass hands out the oracle's gradient tensor by tensor in the order the real one
    finishes them: stage 0 the readout's tensors, stage k block L-k, stage L+1 embedding and encoders."""

    def __init__(self, true_grad, cfg):
        self.group, self.overlap_allreduce, self._pending = None, True, []
        self.true_grad, self.grad = true_grad, torch.zeros_like(true_grad)
        self.theta = torch.zeros_like(true_grad)
        L = cfg.n_layers
        self.dyn = SimpleNamespace(_cfg={'n_layers': L})
        self.offsets, self.stage_of, off = {}, [], 0
        for name, (fo, fi, has_bias) in linear_specs(cfg).items():
            n = fo * fi + (fo if has_bias else 0)
            self.offsets[name + '.weight'] = (off, fo * fi)
            if name.startswith('egnn.e_block_'):
                stage = L - int(name.split('_')[2].split('.')[0])
            elif name.split('.')[0] in ('phar_decoder', 'residue_decoder') or name == 'egnn.embedding_out':
                stage = 0
            else:
                stage = L + 1
            self.stage_of.append((stage, off, off + n))
            off += n
        assert off == true_grad.numel()
        self.calls = []
        outer = self

        class FakeHandle:
            def param_offset(self, name):
                return outer.offsets[name]

            def train_backward_stages(self, d_eps, grad, first, last, d_eps_q=None):
                outer.calls.append((first, last))
                for stage, lo, hi in outer.stage_of:
                    if first <= stage <= last:
                        grad[lo:hi] = outer.true_grad[lo:hi]

            def train_backward(self, d_eps, grad, d_eps_q=None):
                grad.copy_(outer.true_grad)
        self.h = FakeHandle()


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    from helpers import load_golden, loss_case
    cfg = loss_case(load_golden('g6_loss.npz'))[0]
    grad = flat_grad_of_subbatch([0, 1] if rank == 0 else [2, 3])
    stub = StagedStub(grad, cfg)
    chunks = stub.grad_chunks()
    stub._backward(None, None)                        # staged backward, one async all-reduce per finished chunk
    n_async = len(stub._pending)
    stub._allreduce()                                 # wait + average
    plain = StagedStub(grad, cfg)
    plain.overlap_allreduce = False
    plain._backward(None, None)
    plain._allreduce()                                # the single flat bucket
    if rank == 0:
        q.put((stub.grad.numpy(), plain.grad.numpy(), chunks, stub.calls, n_async))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_average_equals_full_batch_gradient():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29650 + os.getpid() % 200
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, got_plain, chunks, calls, n_async = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # the chunks tile the flat buffer back to front and their stage ranges are consecutive: 0..L+1
    assert chunks[0][2] == len(got) and chunks[-1][1] == 0 and all(a[1] == b[2] for a, b in zip(chunks, chunks[1:]))
    assert calls[0][0] == 0 and all(b[0] == a[1] + 1 for a, b in zip(calls, calls[1:])) and n_async == len(chunks) == len(calls)
    assert np.array_equal(got, got_plain)             # overlapped chunks == one flat all-reduce, bit for bit
    want = flat_grad_of_subbatch([0, 1, 2, 3]).numpy()
    assert got.shape == want.shape and np.abs(want).max() > 0
    assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max()
    # and the flat order is the reference's state_dict order: G11's per-tensor gradients line up with it
    from helpers import load_golden
    g = load_golden('g11_train.npz')
    names = [k[len('grad/dynamics.'):] for k in g if k.startswith('grad/dynamics.')]
    flat = np.concatenate([g['grad/dynamics.' + n].reshape(-1) for n in names])
    assert np.abs(flat - want).max() <= 2e-5 * np.abs(want).max()
