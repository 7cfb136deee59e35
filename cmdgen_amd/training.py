"""The training step of PharPocketDDPM on the HIP training path.

Counterpart of ``training_step`` (lightning_modules.py:245-260), ``configure_optimizers`` (:141-143,
AdamW amsgrad, weight_decay 1e-12), ``configure_gradient_clipping`` (:543-568, adaptive norm clipping from a queue
of recent gradient norms) and the DDP gradient averaging Lightning sets up (train.py:111-121).

Design: every trainable tensor of ``EGNNDynamics`` lives in ONE flat fp32 device buffer ``theta`` (the module's
parameters are re-pointed to views of it, so ``state_dict`` / checkpoints / sampling keep working with no copies);
``grad`` and the three AdamW moment buffers have the same layout.  One step =
  loss terms on the activation-saving forward (``cmdgen_train_forward``)  ->  analytic dL/d eps (a few torch ops on
  device)  ->  ``cmdgen_train_backward`` (parameter gradients, in stages)  ->  ``all_reduce`` of the flat gradient over
  RCCL when ``world_size > 1``, three contiguous chunks started as their stages finish  ->  gradient norm, clipping coefficient  ->  ``cmdgen_adamw_step``.
Both objectives ('l2' of the shipped configs, 'vlb' with a predefined schedule) and all model variants train.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from . import utils


def per_sample_table(gamma, log_pn, T, n_dims, norm_values, t_int, n_phar, n_pocket, variant: str = 'conditional') -> torch.Tensor:
    """[12, B] fp32 table of everything ConditionalDDPM.forward derives from t and the node counts alone
    (conditional_model.py:206-221, :49-59; en_diffusion.py:227-234), made on the host in numpy with the reference's fp32 op
    sequence - the `tab` argument of cmdgen_train_noise / cmdgen_train_loss (rows as listed in include/cmdgen_hip.h):
    alpha_t, sigma_t, t_is_zero, SNR weight, alpha_T, sigma_T, -log_constants_p_x_given_z0, delta_log_px, log p(N), t_int, t,
    sigma_t * norm_values[1].  gamma: the predefined schedule's lookup table [T+1]; log_pn: log p(n_phar | n_pocket) table.
    variant: 'conditional' ((n_phar - 1) * n_dims degrees of freedom), 'simple' (SimpleConditionalDDPM, conditional_model.py:481-525:
    n_phar * n_dims) or 'joint' (EnVariationalDiffusion, en_diffusion.py:332-465: (n_phar + n_pocket - 1) * n_dims, and `log_pn` is the
    joint table log p(n_phar, n_pocket))."""
    g, f32 = np.asarray(gamma, dtype=np.float32), np.float32
    Tf = f32(T)
    t_int = np.asarray(torch.as_tensor(t_int).detach().to('cpu', torch.float32)).reshape(-1)
    s, t = (t_int - f32(1)) / Tf, t_int / Tf
    gamma_s, gamma_t = g[np.rint(s * Tf).astype(np.int64)], g[np.rint(t * Tf).astype(np.int64)]     # s = -1/T wraps to gamma[T], as the reference's lookup
    gamma_T, gamma_0 = g[int(round(float(Tf)))], g[0]
    sigmoid = lambda x: (f32(1) / (f32(1) + np.exp(-x, dtype=f32))).astype(f32)
    n = np.asarray(n_phar, dtype=f32)
    if variant == 'joint':
        sub = (n + np.asarray(n_pocket, dtype=f32) - f32(1)) * f32(n_dims)
    else:
        sub = (n if variant == 'simple' else n - f32(1)) * f32(n_dims)
    nv0, nv1 = float(norm_values[0]), float(norm_values[1])
    sigma_t = np.sqrt(sigmoid(gamma_t))
    one = np.ones_like(n)
    tab = np.stack([
        np.sqrt(sigmoid(-gamma_t)), sigma_t, (t_int == 0).astype(f32), f32(1) - np.exp(-(gamma_s - gamma_t), dtype=f32),
        np.sqrt(sigmoid(-gamma_T)) * one, np.sqrt(sigmoid(gamma_T)) * one,
        -(sub * f32(-(0.5 * gamma_0) - 0.5 * np.log(2 * np.pi))), -sub * f32(np.log(nv0)),
        np.asarray(log_pn, dtype=f32)[np.asarray(n_phar, dtype=np.int64), np.asarray(n_pocket, dtype=np.int64)],
        t_int, t, sigma_t * f32(nv1)]).astype(f32)
    return torch.from_numpy(np.ascontiguousarray(tab))


from .collectives import wait_collective  # noqa: E402,F401  (re-exported: bench.py, tools/bench_train.py)

class HipTrainer:
    def __init__(self, model, lr: Optional[float] = None, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-12, clip_grad: Optional[bool] = None, process_group=None,
                 gemm_dtype: str = 'fp32'):
        self.joint = model.mode == 'joint'
        assert model.loss_type in ('l2', 'vlb')
        if getattr(model.ddpm, 'learned_schedule', False):
            raise NotImplementedError("noise_schedule='learned': the training step differentiates a predefined schedule only (sampling works)")
        self.model = model
        self.ddpm = model.ddpm
        self.dyn = model.ddpm.dynamics
        self.lr = float(model.lr if lr is None else lr)
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.clip_grad = bool(getattr(model, 'clip_grad', True) if clip_grad is None else clip_grad)
        self.gradnorm_queue = utils.Queue()
        self.gradnorm_queue.add(3000)                      # large value that will be flushed (lightning_modules.py:78-80)
        self.group = process_group
        self.step_count = 0
        h = self.dyn.hip_handle()
        self.h = h
        n = h.param_count()
        dev = next(self.dyn.parameters()).device
        self.theta = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros_like(self.theta)
        self.exp_avg = torch.zeros_like(self.theta)
        self.exp_avg_sq = torch.zeros_like(self.theta)
        self.max_exp_avg_sq = torch.zeros_like(self.theta)
        # re-point the module's parameters at views of the flat buffer (zero-copy; names as in the state_dict)
        seen = 0
        for name, p in self.dyn.named_parameters():
            off, cnt = h.param_offset(name)
            assert cnt == p.numel(), name
            view = self.theta[off:off + cnt].view(p.shape)
            view.copy_(p.data)
            p.data = view
            seen += cnt
        assert seen <= n < seen + 4 * (len(list(self.dyn.parameters())) + 1), (seen, n)     # + alignment padding
        assert gemm_dtype in ('fp32', 'bf16')
        self.gemm_dtype = gemm_dtype        # 'bf16': GEMM operands in bf16, fp32 accumulation (mixed precision); default exact fp32
        self.last_info: Dict[str, float] = {}
        self.overlap_allreduce = True       # all-reduce finished gradient chunks behind the rest of the backward pass
        self.force_collectives = False      # run the data-parallel path (staged backward, chunked all-reduce) even in a ONE-rank process group (RCCL first contact on a one-GPU box)
        self.fused_loss = True              # noising and loss terms as three library launches (cmdgen_train_noise / _loss[_joint])
        self._gamma_host = self._logpn_host = None
        self._last_fused = None
        self._net_inputs = None
        self.pipelined = False              # True: a step does not wait for its own gradient norm (see optimizer_step)
        self._norm_pending = None
        self.last_grad_norm = None
        self._tab_pinned, self._tab_slot = [None, None], 0
        self._pending = []

    # ------------------------------------------------------------------
    def _net(self, z_t, xh_pocket, t, phar_mask, pocket_mask):
        # cmdgen_train_forward keeps the raw device pointers of its two inputs: the backward pass reads their feature columns
        # for the weight gradients of phar_encoder.0 / residue_encoder.0.  Hold the exact tensors handed over (possibly the
        # temporaries .to().contiguous() made) until the pass has run - otherwise the caching allocator may give their
        # blocks to the loss side's temporaries in between.
        z, q = z_t.to(torch.float32).contiguous(), xh_pocket.to(torch.float32).contiguous()
        self._net_inputs = (z, q)
        out = self.h.train_forward(self.theta, z, q, t, want_pocket=self.joint)
        return out if self.joint else (out, None)

    # ------------------------------------------------------------------ fused loss side
    def _variant(self) -> Optional[str]:
        from .equivariant_diffusion.en_diffusion import EnVariationalDiffusion
        from .equivariant_diffusion.conditional_model import ConditionalDDPM, SimpleConditionalDDPM
        return {ConditionalDDPM: 'conditional', SimpleConditionalDDPM: 'simple', EnVariationalDiffusion: 'joint'}.get(type(self.ddpm))

    def _fused_ok(self) -> bool:
        from .equivariant_diffusion.en_diffusion import PredefinedNoiseSchedule
        return self.fused_loss and self._variant() is not None and isinstance(self.ddpm.gamma, PredefinedNoiseSchedule)

    def _sample_table(self, t_int, n_phar, n_pocket):
        """[TT_COLS, B] per-sample scalars of one step (include/cmdgen_hip.h), see ``per_sample_table``."""
        ddpm, variant = self.ddpm, self._variant()
        if self._gamma_host is None:
            self._gamma_host = ddpm.gamma.gamma.detach().to('cpu', torch.float32).numpy().copy()
            sd = ddpm.size_distribution
            lp = sd.m.logits.view(sd.prob.shape) if variant == 'joint' else sd._table(1, torch.device('cpu'))     # log_pN (:297-299) / log p(n1 | n2)
            self._logpn_host = lp.detach().to('cpu', torch.float32).numpy().copy()
        return per_sample_table(self._gamma_host, self._logpn_host, ddpm.T, ddpm.n_dims, ddpm.norm_values, t_int, n_phar, n_pocket, variant)

    @torch.no_grad()
    def _loss_and_grad_fused(self, data, t_int=None, eps=None):
        """The training loss with the three fused launches of the library (cmdgen_train_noise / cmdgen_train_loss, or their _joint
        forms) around the activation-saving forward; same values as PharPocketDDPM.forward(training mode) for ConditionalDDPM,
        SimpleConditionalDDPM and the joint EnVariationalDiffusion."""
        model, ddpm, h = self.model, self.ddpm, self.h
        model.train()
        dev = self.theta.device
        f32 = lambda k: data[k].to(dev, torch.float32).contiguous()
        px, poh, qx, qoh = f32('phar_coords'), f32('phar_one_hot'), f32('pocket_c_alpha'), f32('pocket_one_hot')
        # node counts: taken from the batch's host copies when the loader kept them (keys '<name>_cpu': no wait on the device,
        # so the previous step's backward pass can still be running), else one device-to-host copy for both
        if 'num_phar_atoms_cpu' in data:
            n_l = np.ascontiguousarray(data['num_phar_atoms_cpu'].numpy().astype(np.int64))
            n_p = np.ascontiguousarray(data['num_pocket_nodes_cpu'].numpy().astype(np.int64))
        else:
            sizes = torch.stack([data['num_phar_atoms'].reshape(-1), data['num_pocket_nodes'].reshape(-1)]).detach().to('cpu', torch.int64).numpy()
            n_l, n_p = np.ascontiguousarray(sizes[0]), np.ascontiguousarray(sizes[1])
        B = len(n_l)
        h.set_layout(n_l, n_p, on_stream=self.pipelined)
        h.train_set_precision(self.gemm_dtype == 'bf16')
        if t_int is None:
            t_int = torch.randint(0, ddpm.T + 1, size=(B, 1)).float()          # training mode: t = 0 included
        tab = self._sample_table(t_int, n_l, n_p)
        if self.pipelined:      # pinned staging, two buffers: the copy is stream-ordered and never waits for the previous step
            self._tab_slot ^= 1
            pin = self._tab_pinned[self._tab_slot]
            if pin is None or pin.shape != tab.shape:
                pin = self._tab_pinned[self._tab_slot] = torch.empty(tab.shape, dtype=torch.float32).pin_memory()
            pin.copy_(tab)
            tab = pin.to(dev, non_blocking=True)
        else:
            tab = tab.to(dev, non_blocking=True)
        tm = tab.mean(1)
        info = {'SNR_weight': tm[3], 'delta_log_px': tm[7], 'neg_log_const_0': tm[6], 'log_pN': tm[8]}
        if self.joint:
            if eps is None:
                e_l = torch.randn((px.shape[0], ddpm.n_dims + ddpm.phar_nf), device=dev)
                e_q = torch.randn((qx.shape[0], ddpm.n_dims + ddpm.residue_nf), device=dev)
            else:
                e_l, e_q = (v.to(dev, torch.float32).contiguous() for v in next(iter(eps)))
            z_t, z_q, e_l, e_q, kl = h.train_noise_joint(px, poh, qx, qoh, tab, e_l, e_q)
            self._net_inputs = (z_t, z_q)
            net_out, net_q = h.train_forward(self.theta, z_t, z_q, tab[10], want_pocket=True)
            terms, means, d_eps, d_eps_q = h.train_loss_joint(model.loss_type == 'l2', float(ddpm.T), net_out, net_q, e_l, e_q, z_t, z_q,
                                                              poh, qoh, tab, kl)
            self.grad.zero_()
            self._backward(d_eps, d_eps_q)
            info.update({'eps_hat_phar_x': means[4], 'eps_hat_phar_h': means[5], 'eps_hat_pocket_x': means[10], 'eps_hat_pocket_h': means[11],
                         'error_t_phar': means[1], 'error_t_pocket': means[9], 'loss_0': means[2], 'kl_prior': means[3]})
            self._last_fused = {'terms': terms, 'tab': tab, 'z_t': z_t, 'xh_pocket': z_q, 'eps_t': e_l, 'eps_t_pocket': e_q, 'net_out': net_out,
                                'net_out_pocket': net_q, 'd_eps': d_eps, 'd_eps_pocket': d_eps_q}
            self._net_inputs = None
            return means[0], terms[:, 0], info
        if eps is None:
            e = torch.randn((px.shape[0], ddpm.n_dims + ddpm.phar_nf), device=dev)
        else:
            e = next(iter(eps)).to(dev, torch.float32).contiguous()
        z_t, xh_pocket, kl = h.train_noise(px, poh, qx, qoh, tab, e)
        net_out = h.train_forward(self.theta, z_t, xh_pocket, tab[10])
        terms, means, d_eps = h.train_loss(model.loss_type == 'l2', float(ddpm.T), net_out, e, z_t, poh, tab, kl)
        self.grad.zero_()
        self._backward(d_eps, None)
        info.update({'eps_hat_phar_x': means[4], 'eps_hat_phar_h': means[5], 'error_t_phar': means[1],
                     'error_t_pocket': torch.zeros((), device=dev), 'loss_0': means[2], 'kl_prior': means[3]})
        self._last_fused = {'terms': terms, 'tab': tab, 'z_t': z_t, 'xh_pocket': xh_pocket, 'eps_t': e, 'net_out': net_out, 'd_eps': d_eps}
        return means[0], terms[:, 0], info

    @torch.no_grad()
    def loss_and_grad(self, data, t_int=None, eps=None):
        """-> (loss, nll [B], info); leaves dL/d theta (this rank's batch mean) in ``self.grad``."""
        if self._fused_ok():
            return self._loss_and_grad_fused(data, t_int=t_int, eps=eps)
        model = self.model
        model.train()
        phar, pocket = model.get_phar_and_pocket(data)
        B = len(phar['size'])
        self.h.set_layout(phar['size'].detach().to('cpu', torch.int64).numpy(),
                          pocket['size'].detach().to('cpu', torch.int64).numpy())
        self.h.train_set_precision(self.gemm_dtype == 'bf16')
        nll, info = model.forward(data, t_int=t_int, eps=eps, _net=self._net)
        loss = nll.mean(0)
        ctx = self.ddpm._last_train_ctx
        eps_t, net_out, t_is_zero = ctx['eps_t'], ctx['net_out'], ctx['t_is_zero'].squeeze(1)
        nd, pnf = self.ddpm.n_dims, self.ddpm.phar_nf
        n_b = phar['size'].to(torch.float32)
        # d loss / d net_out (lightning_modules.py:198-217; conditional_model.py:243-262, :291-301):
        #   l2 : loss = mean_b [ 0.5 * error_t/((nd+P) n_b) * (t != 0) + loss_0_x/(nd n_b) * (t == 0) + const ]
        #   vlb: loss = mean_b [ -T/2 * SNR_weight_b * error_t * (t != 0) + loss_0_x * (t == 0) + const ]
        # with error_t = sum (eps - net)^2 and loss_0_x = 0.5 * sum over x of (eps - net)^2
        diff = net_out - eps_t
        l2 = model.loss_type == 'l2'
        if l2:
            s_t, s_0 = 1.0 / ((nd + pnf) * n_b), 1.0 / (nd * n_b)
        else:
            s_t, s_0 = -self.ddpm.T * ctx['SNR_weight'], torch.ones_like(n_b)
        w_t = ((1.0 - t_is_zero) * s_t / B)[phar['mask']]
        w_0 = (t_is_zero * s_0 / B)[phar['mask']]
        d_eps = diff * w_t[:, None]
        d_eps[:, :nd] += diff[:, :nd] * w_0[:, None]
        d_eps_q = None
        if self.joint:      # the pocket is generated too: the same two terms on its nodes (lightning_modules.py:201-208)
            rnf, n_q = self.ddpm.residue_nf, pocket['size'].to(torch.float32)
            diff_q = ctx['net_out_pocket'] - ctx['eps_t_pocket']
            q_t, q_0 = (1.0 / ((nd + rnf) * n_q), 1.0 / (nd * n_q)) if l2 else (s_t, s_0)
            d_eps_q = diff_q * ((1.0 - t_is_zero) * q_t / B)[pocket['mask']][:, None]
            d_eps_q[:, :nd] += diff_q[:, :nd] * (t_is_zero * q_0 / B)[pocket['mask']][:, None]
            d_eps_q = d_eps_q.contiguous()
        self.grad.zero_()
        self._backward(d_eps.contiguous(), d_eps_q)
        self._net_inputs = None         # (stream-ordered allocator: blocks freed now are reused only behind the queued pass)
        return loss, nll, info

    # ------------------------------------------------------------------ data parallelism
    def _world(self) -> int:
        import torch.distributed as dist
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def _dp_active(self) -> bool:
        """Does the step go through the collectives?  (more than one rank - or a one-rank group on request)"""
        import torch.distributed as dist
        return self._world() > 1 or (self.force_collectives and dist.is_available() and dist.is_initialized())

    def grad_chunks(self):
        """The flat gradient as the contiguous ranges that become final one after the other during the backward pass:
        [(last stage of the pass that completes the range, lo, hi)], back of the buffer first (DDP's reverse-order
        buckets, train.py:111-121): the upper half of the blocks, the lower half, then the small head (encoders,
        decoders, embeddings) that is only complete when the whole pass is."""
        L = int(self.dyn._cfg['n_layers'])
        n = self.theta.numel()
        start = lambda l: self.h.param_offset(f'egnn.e_block_{l}.gcl_0.edge_mlp.0.weight')[0]
        mid = L // 2
        chunks = []
        if L - mid > 0:
            chunks.append((L - mid, start(mid), n))              # stages 0..L-mid: readout + blocks L-1..mid
        if mid > 0:
            chunks.append((L, start(0), start(mid)))             # blocks mid-1..0
        chunks.append((L + 1, 0, start(0)))                      # embedding / encoders (and the readout's tensors)
        return chunks

    def _backward(self, d_eps, d_eps_q):
        """Backward pass; with several ranks the all-reduce of every finished chunk is started as soon as its stages
        are queued, so the collective runs behind the remaining differentiation (RCCL on its own stream) and only the
        small head chunk is exposed.  The sum is divided by the world size in ``_allreduce``."""
        self._pending = []
        if not self._dp_active() or not self.overlap_allreduce:
            self.h.train_backward(d_eps, self.grad, d_eps_q)
            return
        import torch.distributed as dist
        first = 0
        for last, lo, hi in self.grad_chunks():
            self.h.train_backward_stages(d_eps, self.grad, first, last, d_eps_q)
            self._pending.append(dist.all_reduce(self.grad[lo:hi], group=self.group, async_op=True))
            first = last + 1

    def _allreduce(self):
        world = self._world()
        if self._dp_active():
            import torch.distributed as dist
            if self._pending:
                for work in self._pending:
                    work.wait()
                self._pending = []
            else:
                wait_collective(dist.all_reduce(self.grad, group=self.group, async_op=True))          # one flat bucket over RCCL
            self.grad.div_(world)

    def broadcast_state(self, src: int = 0):
        """Every replica starts from rank `src`'s parameters and optimizer state (what DDP does at construction,
        train.py:117-118): replicas built from differently seeded RNGs would otherwise drift apart silently, since
        only gradients are averaged."""
        if self._dp_active():
            import torch.distributed as dist
            for t in (self.theta, self.exp_avg, self.exp_avg_sq, self.max_exp_avg_sq):
                wait_collective(dist.broadcast(t, src=src, group=self.group, async_op=True))
            self.dyn._weights_sig = None

    def _collect_norm(self):
        """The deferred gradient norm of the previous step enters the queue of recent norms (pipelined mode)."""
        if self._norm_pending is not None:
            mx = self._norm_pending
            self._norm_pending = None
            grad_norm = self.h.last_grad_norm()
            self.last_grad_norm = grad_norm
            if self._left_half_range(grad_norm):
                import warnings
                warnings.warn('the previous training step produced a non-finite gradient on the half matrix engine (an activation beyond fp16\'s 65504): '
                              'its update was skipped on the device; the forward runs on the bf16 split engine from here on', RuntimeWarning, stacklevel=3)
                return
            if self.clip_grad:
                self.gradnorm_queue.add(float(mx) if grad_norm > mx else grad_norm)
                if grad_norm > mx:
                    print(f'Clipped gradient with value {grad_norm:.1f} while allowed {mx:.1f}')

    def _left_half_range(self, grad_norm) -> bool:
        """A non-finite gradient norm after a forward on the half matrix engine (two fp16 pieces per operand: range 65504): the device has skipped
        the update (k_adamw); switch this handle's training forward to the three-piece bf16 engine (fp32's range).  True when that happened."""
        import math
        if grad_norm is None or math.isfinite(grad_norm):
            return False
        if self.h.get_option('train_half') == 0 or not self.h.half_engine_active():
            return False                    # not the half engine's doing: the reference would carry the NaN on as well
        self.h.set_option('train_half', 0)
        self.half_range_fallbacks = getattr(self, 'half_range_fallbacks', 0) + 1
        return True

    def optimizer_step(self, max_grad_norm: Optional[float] = None):
        """Adaptive clipping + AdamW(amsgrad) on the flat buffers; returns (grad_norm, max_grad_norm).  With
        ``self.pipelined`` the norm is NOT waited for (returned as None, collected before the next step's bound is formed:
        ``last_grad_norm`` then holds it), so the host goes on to queue the next step behind this one."""
        self._collect_norm()
        self.step_count += 1
        if self.clip_grad and max_grad_norm is None:                   # 150 % of the recent mean + 2 stdev
            max_grad_norm = 1.5 * self.gradnorm_queue.mean() + 2 * self.gradnorm_queue.std()
        # norm -> clipping coefficient -> update are queued back to back (the coefficient is formed on the device from the
        # bound known beforehand); the norm itself is only read back for the queue of recent norms
        grad_norm = self.h.adamw_step_clipped(self.theta, self.grad, self.exp_avg, self.exp_avg_sq, self.max_exp_avg_sq,
                                              self.step_count, self.lr, self.betas, self.eps, self.weight_decay,
                                              float(max_grad_norm) if self.clip_grad else 0.0, defer=self.pipelined)
        self.dyn._weights_sig = None            # the sampler's packed copy of the weights is stale now
        if self.pipelined:
            self._norm_pending = float(max_grad_norm) if self.clip_grad else float('inf')
            return None, max_grad_norm
        self.last_grad_norm = grad_norm
        if self._left_half_range(grad_norm):
            self.step_count -= 1            # the device skipped this update; training_step repeats the batch on the bf16 engine
            return grad_norm, max_grad_norm
        if self.clip_grad:
            self.gradnorm_queue.add(float(max_grad_norm) if grad_norm > max_grad_norm else grad_norm)
            if grad_norm > max_grad_norm:
                print(f'Clipped gradient with value {grad_norm:.1f} while allowed {max_grad_norm:.1f}')
        return grad_norm, max_grad_norm

    def training_step(self, data, t_int=None, eps=None, max_grad_norm: Optional[float] = None):
        loss, nll, info = self.loss_and_grad(data, t_int=t_int, eps=eps)
        self._allreduce()
        before = getattr(self, 'half_range_fallbacks', 0)
        grad_norm, mx = self.optimizer_step(max_grad_norm)
        if getattr(self, 'half_range_fallbacks', 0) != before and not self.pipelined:
            import warnings
            warnings.warn('non-finite gradient on the half matrix engine (an activation beyond fp16\'s 65504): the step is repeated on the bf16 split engine',
                          RuntimeWarning, stacklevel=2)
            loss, nll, info = self.loss_and_grad(data, t_int=t_int, eps=eps)
            self._allreduce()
            grad_norm, mx = self.optimizer_step(max_grad_norm)
        info = dict(info)
        info['loss'] = loss
        info['grad_norm'] = grad_norm
        self.last_info = info
        return info
