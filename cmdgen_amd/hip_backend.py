"""ctypes binding of libcmdgen_hip.so (include/cmdgen_hip.h).

There is no CPU fallback: if the library has not been built
(``python __graft_entry__.py``) or no MI355X is visible, using the model raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Sequence

import numpy as np

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libcmdgen_hip.so')
_lib = None


class CmdgenError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        'phar_nf', 'residue_nf', 'joint_nf', 'hidden_nf', 'n_layers', 'inv_sublayers',
        'attention', 'tanh', 'condition_time', 'timesteps', 'no_com_projection', 'update_pocket_coords')] + \
        [(n, C.c_float) for n in ('edge_cutoff', 'norm_constant', 'normalization_factor',
                                  'coords_range', 'norm_x', 'norm_h', 'bias_h')] + [('aggregation_mean', C.c_int32), ('sin_embedding', C.c_int32)]


class Counters(C.Structure):
    _fields_ = [('evaluations', C.c_uint64), ('edges', C.c_uint64), ('edges_phar', C.c_uint64),
                ('nodes', C.c_uint64), ('nan_resets', C.c_uint64), ('edges_skipped', C.c_uint64), ('node_rows_skipped', C.c_uint64),
                ('reserved', C.c_uint64 * 1)]


class KernelTimes(C.Structure):
    _fields_ = [(n, C.c_float) for n in ('edge_build_ms', 'embed_ms', 'edge_msg_ms', 'node_ms',
                                         'edge_coord_ms', 'readout_ms', 'ddpm_ms')] + \
               [(n, C.c_int32) for n in ('edge_msg_launches', 'node_launches', 'edge_coord_launches')]


# every symbol include/cmdgen_hip.h declares: (name, restype, argtypes)
_vp, _fp, _i64p = C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)
SYMBOLS = [
    ('cmdgen_create', C.c_int, [C.POINTER(Config), C.c_int, C.POINTER(_vp)]),
    ('cmdgen_destroy', None, [_vp]),
    ('cmdgen_last_error', C.c_char_p, [_vp]),
    ('cmdgen_version', C.c_char_p, []),
    ('cmdgen_load_weights', C.c_int, [_vp, C.c_char_p, _vp, C.c_size_t]),
    ('cmdgen_finalize_weights', C.c_int, [_vp]),
    ('cmdgen_set_layout', C.c_int, [_vp, C.c_int64, _i64p, _i64p]),
    ('cmdgen_set_layout_on_stream', C.c_int, [_vp, C.c_int64, _i64p, _i64p, _vp]),
    ('cmdgen_dynamics_forward', C.c_int, [_vp, _fp, _fp, _fp, _fp, _fp, _vp]),
    ('cmdgen_get_edges', C.c_int, [_vp, _vp, _vp, C.c_int64, _i64p, _vp]),
    ('cmdgen_debug_read', C.c_int, [_vp, C.c_char_p, _vp, C.c_size_t, _vp]),
    ('cmdgen_debug_eval_prefix', C.c_int, [_vp, _fp, _fp, _fp, C.c_int32, C.c_int32, _vp]),
    ('cmdgen_radius_graph', C.c_int, [_vp, _fp, _i64p, C.c_int64, _vp, _vp, C.c_int64, _i64p, _vp]),
    ('cmdgen_debug_noise', C.c_int, [_vp, C.c_uint64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _fp, _vp]),
    ('cmdgen_sample_chain', C.c_int, [_vp, _fp, _fp, C.c_int32, _fp, C.c_uint64, _i64p, _fp, _fp, _fp, _fp,
                                      C.c_int32, _vp]),
    ('cmdgen_joint_chain', C.c_int, [_vp, _fp, _fp, _fp, _fp, _fp, _fp, C.c_int32, C.c_int32, C.c_int32, _fp, C.c_int64,
                                     C.c_uint64, _i64p, _fp, _fp, _fp, C.c_int32, _vp]),
    ('cmdgen_joint_plan', C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _i64p, _i64p]),
    ('cmdgen_param_count', C.c_int, [_vp, _i64p]),
    ('cmdgen_param_offset', C.c_int, [_vp, C.c_char_p, _i64p, _i64p]),
    ('cmdgen_train_forward', C.c_int, [_vp, _fp, _fp, _fp, _fp, _fp, _fp, _vp]),
    ('cmdgen_train_backward', C.c_int, [_vp, _fp, _fp, _fp, _vp]),
    ('cmdgen_train_backward_stages', C.c_int, [_vp, _fp, _fp, _fp, C.c_int32, C.c_int32, _vp]),
    ('cmdgen_train_set_precision', C.c_int, [_vp, C.c_int32]),
    ('cmdgen_train_noise', C.c_int, [_vp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _vp]),
    ('cmdgen_train_loss', C.c_int, [_vp, C.c_int32, C.c_float, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _vp]),
    ('cmdgen_train_noise_joint', C.c_int, [_vp] + [_fp] * 12 + [_vp]),
    ('cmdgen_train_loss_joint', C.c_int, [_vp, C.c_int32, C.c_float] + [_fp] * 14 + [_vp]),
    ('cmdgen_grad_sqnorm', C.c_int, [_vp, _fp, C.c_int64, C.POINTER(C.c_float), _vp]),
    ('cmdgen_adamw_step', C.c_int, [_vp, _fp, _fp, _fp, _fp, _fp, C.c_int64, C.c_int64, C.c_float, C.c_float, C.c_float,
                                    C.c_float, C.c_float, C.c_float, _vp]),
    ('cmdgen_adamw_step_clipped', C.c_int, [_vp, _fp, _fp, _fp, _fp, _fp, C.c_int64, C.c_int64, C.c_float, C.c_float, C.c_float,
                                            C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float), _vp]),
    ('cmdgen_last_grad_norm', C.c_int, [_vp, C.POINTER(C.c_float)]),
    ('cmdgen_debug_wgrad', C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, _fp, _fp, _fp, _fp, C.c_int32, _vp]),
    ('cmdgen_debug_dgrad', C.c_int, [_vp, C.c_int32, _fp, _fp, _fp, _fp, _fp, C.c_int32, C.c_float, _fp, C.c_int32, C.c_int32, _vp]),
    ('cmdgen_debug_sgemm', C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _fp, C.c_int32, _fp,
                                     C.c_int32, _fp, C.c_int32, _fp, C.c_int32, C.c_int32, _vp]),
    ('cmdgen_set_step_table', C.c_int, [_vp, C.c_int32, _vp]),
    ('cmdgen_chain_status', C.c_int, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float), _i64p, _vp]),
    ('cmdgen_get_counters', C.c_int, [_vp, C.POINTER(Counters), _vp]),
    ('cmdgen_reset_counters', C.c_int, [_vp, _vp]),
    ('cmdgen_profile_evaluation', C.c_int, [_vp, _fp, _fp, _fp, _fp, C.POINTER(KernelTimes), _vp]),
    ('cmdgen_query', C.c_int, [_vp, C.c_char_p, _i64p]),
    ('cmdgen_set_gemm_mode', C.c_int, [_vp, C.c_int32]),
    ('cmdgen_set_option', C.c_int, [_vp, C.c_char_p, C.c_int64, C.c_int32]),
    ('cmdgen_get_option', C.c_int, [_vp, C.c_char_p, _i64p, C.POINTER(C.c_int32)]),
    ('cmdgen_debug_stamps', C.c_int, [_vp, C.POINTER(C.c_uint64), C.c_int32]),
    ('cmdgen_time_evaluation', C.c_int, [_vp, _fp, _fp, _fp, _fp, C.c_int32, C.c_int32, C.POINTER(C.c_float), _vp]),
    ('cmdgen_time_edge_kernel', C.c_int, [_vp, C.c_int32, C.c_int32, C.POINTER(C.c_float), _vp]),
    ('cmdgen_set_kernel_profiling', C.c_int, [_vp, C.c_int32]),
    ('cmdgen_get_kernel_profile', C.c_int, [_vp, C.POINTER(C.c_float), _i64p, _vp]),
]


def library_path() -> str:
    return _LIB_PATH


def load_library():
    """dlopen the in-tree library and bind every declared symbol (no device needed)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get('CMDGEN_LIB', _LIB_PATH)      # diagnostic builds (-DCMDGEN_STAMPS=n, tools/build_variant.sh) only
    if not os.path.exists(path):
        raise CmdgenError(
            f'{path} is missing: build it with `python __graft_entry__.py` '
            '(hipcc --offload-arch=gfx950). There is no CPU fallback for this path.')
    import torch  # noqa: F401  - loads the process's HIP runtime first so both share it
    lib = C.CDLL(path)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)          # AttributeError if the .so does not export it
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


# Options every NEW Handle of this process starts with (cmdgen_set_option; the library itself reads no environment variable).
# The parity tests pin tile sizes through it (monkeypatch.setitem), tools/ and bench.py fill it from their command line.
DEFAULT_OPTIONS: Dict[str, int] = {}


def parse_options(text: str) -> Dict[str, int]:
    """'edge_mt=128,node64=0' -> {'edge_mt': 128, 'node64': 0}"""
    out = {}
    for item in (text or '').replace(';', ',').split(','):
        item = item.strip()
        if item:
            k, v = item.split('=')
            out[k.strip()] = int(v)
    return out


class Handle:
    """One cmdgen_handle: bound to one device, not thread-safe (as the reference's modules)."""

    def __init__(self, cfg: Dict, device_index: int = 0):
        import torch
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise CmdgenError('no GPU visible: the DiffPhar denoising path runs on MI355X (gfx950) only; '
                              'there is no CPU fallback')
        c = Config()
        c.phar_nf, c.residue_nf = int(cfg['phar_nf']), int(cfg['residue_nf'])
        c.joint_nf, c.hidden_nf, c.n_layers = int(cfg['joint_nf']), int(cfg['hidden_nf']), int(cfg['n_layers'])
        c.inv_sublayers = int(cfg.get('inv_sublayers', 1))
        c.attention, c.tanh = int(bool(cfg['attention'])), int(bool(cfg['tanh']))
        c.condition_time = int(bool(cfg.get('condition_time', True)))
        c.timesteps = int(cfg['timesteps'])
        c.no_com_projection = int(bool(cfg.get('no_com_projection', False)))
        c.update_pocket_coords = int(bool(cfg.get('update_pocket_coords', False)))
        ec = cfg.get('edge_cutoff')
        c.edge_cutoff = -1.0 if ec is None else float(ec)
        c.norm_constant = float(cfg['norm_constant'])
        c.normalization_factor = float(cfg['normalization_factor'])
        c.coords_range = float(cfg.get('coords_range', 15.0))
        nv, nb = cfg['norm_values'], cfg['norm_biases']
        c.norm_x, c.norm_h = float(nv[0]), float(nv[1])
        c.bias_h = float(nb[1] if nb[1] is not None else 0.0)
        if cfg.get('aggregation_method', 'sum') not in ('sum', 'mean'):
            raise CmdgenError("aggregation_method must be 'sum' or 'mean' (egnn_new.py:277-292)")
        c.aggregation_mean = int(cfg.get('aggregation_method', 'sum') == 'mean')
        c.sin_embedding = int(bool(cfg.get('sin_embedding', False)))
        h = C.c_void_p()
        rc = self.lib.cmdgen_create(C.byref(c), int(device_index), C.byref(h))
        if rc != 0:
            raise CmdgenError('cmdgen_create: ' + self.lib.cmdgen_last_error(None).decode())
        self.h = h
        self.device_index = device_index
        self.cfg = dict(cfg)
        self._layout_key = None
        self.n_phar = self.n_pocket = self.batch = 0
        for k, v in DEFAULT_OPTIONS.items():
            self.set_option(k, v)

    # ---- helpers
    def _check(self, rc, what):
        if rc != 0:
            raise CmdgenError(f'{what}: ' + self.lib.cmdgen_last_error(self.h).decode())

    def close(self):
        if getattr(self, 'h', None):
            self.lib.cmdgen_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _stream():
        import torch
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    # ---- parameters
    def load_state_dict(self, state: Dict[str, 'np.ndarray'], prefix: str = 'ddpm.'):
        """state: reference checkpoint names -> arrays/tensors (only the 'ddpm.' sub-tree is used)."""
        for k, v in state.items():
            if not k.startswith(prefix):
                continue
            name = k[len(prefix):]
            if name == 'buffer':
                continue
            a = v.detach().cpu().numpy() if hasattr(v, 'detach') else np.asarray(v)
            a = np.ascontiguousarray(a, dtype=np.float32)
            self._check(self.lib.cmdgen_load_weights(self.h, name.encode(), a.ctypes.data_as(C.c_void_p), a.size),
                        'cmdgen_load_weights')
        self._check(self.lib.cmdgen_finalize_weights(self.h), 'cmdgen_finalize_weights')

    # ---- layout
    def set_layout(self, num_phar: Sequence[int], num_pocket: Sequence[int], on_stream: bool = False):
        """on_stream: order the upload on torch's current stream instead of waiting for the device (cmdgen_set_layout_on_stream)."""
        a = np.ascontiguousarray(np.asarray(num_phar, dtype=np.int64))
        b = np.ascontiguousarray(np.asarray(num_pocket, dtype=np.int64))
        assert a.shape == b.shape and a.ndim == 1
        key = (a.tobytes(), b.tobytes())
        if key != self._layout_key:
            if on_stream:
                self._check(self.lib.cmdgen_set_layout_on_stream(self.h, len(a), a.ctypes.data_as(_i64p), b.ctypes.data_as(_i64p),
                                                                 self._stream()), 'cmdgen_set_layout_on_stream')
            else:
                self._check(self.lib.cmdgen_set_layout(self.h, len(a), a.ctypes.data_as(_i64p), b.ctypes.data_as(_i64p)),
                            'cmdgen_set_layout')
            self._layout_key = key
        self.batch, self.n_phar, self.n_pocket = len(a), int(a.sum()), int(b.sum())

    # ---- one evaluation
    def dynamics_forward(self, xh_phar, xh_pocket, t, want_pocket: bool = True):
        import torch
        P, R = self.cfg['phar_nf'], self.cfg['residue_nf']
        assert xh_phar.is_cuda and xh_phar.dtype == torch.float32 and xh_phar.is_contiguous()
        assert xh_pocket.is_cuda and xh_pocket.dtype == torch.float32 and xh_pocket.is_contiguous()
        assert tuple(xh_phar.shape) == (self.n_phar, 3 + P), (xh_phar.shape, self.n_phar)
        assert tuple(xh_pocket.shape) == (self.n_pocket, 3 + R)
        t = t.reshape(-1).to(torch.float32).contiguous()
        if t.numel() == 1 and self.batch > 1:
            t = t.expand(self.batch).contiguous()
        assert t.numel() == self.batch
        eps_phar = torch.empty_like(xh_phar)
        eps_pocket = torch.empty_like(xh_pocket) if want_pocket else None
        self._check(self.lib.cmdgen_dynamics_forward(self.h, _ptr(xh_phar), _ptr(xh_pocket), _ptr(t),
                                                     _ptr(eps_phar), _ptr(eps_pocket), self._stream()),
                    'cmdgen_dynamics_forward')
        return eps_phar, eps_pocket

    def get_edges(self):
        cap = int(np.sum((np.frombuffer(self._layout_key[0], dtype=np.int64) +
                          np.frombuffer(self._layout_key[1], dtype=np.int64)) ** 2))
        row = np.empty(cap, dtype=np.int32)
        col = np.empty(cap, dtype=np.int32)
        n = C.c_int64(0)
        self._check(self.lib.cmdgen_get_edges(self.h, row.ctypes.data_as(C.c_void_p), col.ctypes.data_as(C.c_void_p),
                                              cap, C.byref(n), self._stream()), 'cmdgen_get_edges')
        return np.stack([row[:n.value], col[:n.value]])

    def radius_graph(self, x, counts):
        """EGNNDynamics.get_edges for arbitrary coordinates: x device [sum counts, 3] (samples back to back),
        counts per sample -> int32 device tensors (row, col) sorted by (row, col)."""
        import torch
        cnt = np.ascontiguousarray(np.asarray(counts, dtype=np.int64))
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and tuple(x.shape) == (int(cnt.sum()), 3)
        cap = int((cnt * cnt).sum())
        row = torch.empty(max(cap, 1), dtype=torch.int32, device=x.device)
        col = torch.empty(max(cap, 1), dtype=torch.int32, device=x.device)
        n = C.c_int64(0)
        self._check(self.lib.cmdgen_radius_graph(self.h, _ptr(x), cnt.ctypes.data_as(_i64p), len(cnt), _ptr(row), _ptr(col),
                                                 cap, C.byref(n), self._stream()), 'cmdgen_radius_graph')
        return row[:n.value], col[:n.value]

    def debug_eval_prefix(self, xh_phar, xh_pocket, t, block: int, stage: int):
        """Run one evaluation up to (block, stage) - stage 1 after the edge-message kernel, 2 after the node kernel,
        3 after the coordinate kernel - and leave the intermediates for debug_read (parity aid)."""
        import torch
        t = t.reshape(-1).to(torch.float32).contiguous()
        if t.numel() == 1 and self.batch > 1:
            t = t.expand(self.batch).contiguous()
        self._check(self.lib.cmdgen_debug_eval_prefix(self.h, _ptr(xh_phar), _ptr(xh_pocket), _ptr(t), int(block), int(stage),
                                                      self._stream()), 'cmdgen_debug_eval_prefix')

    def debug_read(self, what: str, n: int):
        out = np.empty(n, dtype=np.float32)
        self._check(self.lib.cmdgen_debug_read(self.h, what.encode(), out.ctypes.data_as(C.c_void_p), n, self._stream()),
                    'cmdgen_debug_read')
        return out

    # ---- the chain
    def set_step_table(self, K: int, coef: 'np.ndarray'):
        a = np.ascontiguousarray(coef, dtype=np.float32)
        assert a.shape == (K + 1, 4)
        self._check(self.lib.cmdgen_set_step_table(self.h, K, a.ctypes.data_as(C.c_void_p)), 'cmdgen_set_step_table')

    def sample_chain(self, pocket_x, pocket_onehot, timesteps: int, noise=None, seed: int = 0,
                     pocket_ids: Optional[Sequence[int]] = None, want_steps: bool = False,
                     use_graph: bool = True):
        import torch
        P, R = self.cfg['phar_nf'], self.cfg['residue_nf']
        dev = pocket_x.device
        assert pocket_x.is_cuda and pocket_x.dtype == torch.float32 and pocket_x.is_contiguous()
        assert pocket_onehot.dtype == torch.float32 and pocket_onehot.is_contiguous()
        assert tuple(pocket_x.shape) == (self.n_pocket, 3) and tuple(pocket_onehot.shape) == (self.n_pocket, R)
        if noise is not None:
            assert noise.is_cuda and noise.dtype == torch.float32 and noise.is_contiguous()
            assert tuple(noise.shape) == (timesteps + 2, self.n_phar, 3 + P), noise.shape
        xh_phar = torch.empty((self.n_phar, 3 + P), dtype=torch.float32, device=dev)
        xh_pocket = torch.empty((self.n_pocket, 3 + R), dtype=torch.float32, device=dev)
        z_steps = torch.empty((timesteps, self.n_phar, 3 + P), dtype=torch.float32, device=dev) if want_steps else None
        p_steps = torch.empty((timesteps, self.n_pocket, 3), dtype=torch.float32, device=dev) if want_steps else None
        self.last_pocket_steps = p_steps
        ids = None
        if pocket_ids is not None:
            ids = np.ascontiguousarray(np.asarray(pocket_ids, dtype=np.int64))
            assert len(ids) == self.batch
        self._check(self.lib.cmdgen_sample_chain(
            self.h, _ptr(pocket_x), _ptr(pocket_onehot), int(timesteps), _ptr(noise), C.c_uint64(seed & (2 ** 64 - 1)),
            ids.ctypes.data_as(_i64p) if ids is not None else None, _ptr(xh_phar), _ptr(xh_pocket),
            _ptr(z_steps), _ptr(p_steps), int(bool(use_graph)), self._stream()), 'cmdgen_sample_chain')
        return xh_phar, xh_pocket, z_steps

    def joint_plan(self, timesteps: int, resamplings: int = 1, jump_length: int = 1, inpaint: bool = True):
        """(denoising steps, combined noise draws) of a joint chain."""
        a, b = C.c_int64(0), C.c_int64(0)
        self._check(self.lib.cmdgen_joint_plan(self.h, int(timesteps), int(resamplings), int(jump_length),
                                               int(bool(inpaint)), C.byref(a), C.byref(b)), 'cmdgen_joint_plan')
        return a.value, b.value

    def joint_chain(self, timesteps: int, phar=None, pocket=None, phar_fixed=None, pocket_fixed=None,
                    resamplings: int = 1, jump_length: int = 1, noise=None, seed: int = 0,
                    pocket_ids: Optional[Sequence[int]] = None, want_steps: bool = False, use_graph: bool = True,
                    device=None):
        """EnVariationalDiffusion.sample (no fixed masks) / .inpaint.  phar / pocket: (x [n,3], one_hot [n,F]) device
        tensors (inpainting only); *_fixed: float [n] device tensors."""
        import torch
        P, R = self.cfg['phar_nf'], self.cfg['residue_nf']
        inpaint = phar_fixed is not None or pocket_fixed is not None
        row = self.n_phar * (3 + P) + self.n_pocket * (3 + R)
        n_steps, n_draws = self.joint_plan(timesteps, resamplings, jump_length, inpaint)
        ptrs = [None] * 6
        if inpaint:
            ptrs = [phar[0], phar[1], pocket[0], pocket[1], phar_fixed, pocket_fixed]
            shapes = [(self.n_phar, 3), (self.n_phar, P), (self.n_pocket, 3), (self.n_pocket, R), (self.n_phar,), (self.n_pocket,)]
            for t, sh in zip(ptrs, shapes):
                assert t is not None and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == sh, sh
            device = ptrs[0].device
        dev = device if device is not None else torch.device('cuda', self.device_index)
        if noise is not None:
            assert noise.is_cuda and noise.dtype == torch.float32 and noise.is_contiguous()
            assert noise.dim() == 2 and noise.shape[1] == row and noise.shape[0] >= n_draws, (noise.shape, n_draws, row)
        xh_phar = torch.empty((self.n_phar, 3 + P), dtype=torch.float32, device=dev)
        xh_pocket = torch.empty((self.n_pocket, 3 + R), dtype=torch.float32, device=dev)
        z_steps = torch.empty((n_steps, row), dtype=torch.float32, device=dev) if want_steps else None
        ids = None
        if pocket_ids is not None:
            ids = np.ascontiguousarray(np.asarray(pocket_ids, dtype=np.int64))
            assert len(ids) == self.batch
        self._check(self.lib.cmdgen_joint_chain(
            self.h, *[_ptr(t) for t in ptrs], int(timesteps), int(resamplings), int(jump_length), _ptr(noise),
            int(noise.shape[0]) if noise is not None else 0, C.c_uint64(seed & (2 ** 64 - 1)),
            ids.ctypes.data_as(_i64p) if ids is not None else None, _ptr(xh_phar), _ptr(xh_pocket), _ptr(z_steps),
            int(bool(use_graph)), self._stream()), 'cmdgen_joint_chain')
        return xh_phar, xh_pocket, z_steps

    # ---- training step (flat parameter / gradient buffers are torch tensors owned by the caller)
    def param_count(self) -> int:
        n = C.c_int64(0)
        self._check(self.lib.cmdgen_param_count(self.h, C.byref(n)), 'cmdgen_param_count')
        return n.value

    def param_offset(self, name: str):
        a, b = C.c_int64(0), C.c_int64(0)
        self._check(self.lib.cmdgen_param_offset(self.h, name.encode(), C.byref(a), C.byref(b)), 'cmdgen_param_offset')
        return a.value, b.value

    def train_forward(self, theta, xh_phar, xh_pocket, t, want_pocket: bool = False):
        import torch
        assert theta.is_cuda and theta.dtype == torch.float32 and theta.is_contiguous() and theta.numel() == self.param_count()
        P, R = self.cfg['phar_nf'], self.cfg['residue_nf']
        assert tuple(xh_phar.shape) == (self.n_phar, 3 + P) and tuple(xh_pocket.shape) == (self.n_pocket, 3 + R)
        assert xh_phar.is_contiguous() and xh_pocket.is_contiguous() and xh_phar.dtype == torch.float32
        t = t.reshape(-1).to(torch.float32).contiguous()
        assert t.numel() == self.batch
        eps = torch.empty_like(xh_phar)
        eps_q = torch.empty_like(xh_pocket) if want_pocket else None
        self._keep = (theta, xh_phar, xh_pocket, t)          # the backward pass reads them again
        self._check(self.lib.cmdgen_train_forward(self.h, _ptr(theta), _ptr(xh_phar), _ptr(xh_pocket), _ptr(t), _ptr(eps),
                                                  _ptr(eps_q), self._stream()), 'cmdgen_train_forward')
        return (eps, eps_q) if want_pocket else eps

    def train_backward(self, d_eps, grad, d_eps_pocket=None):
        import torch
        assert d_eps.is_cuda and d_eps.dtype == torch.float32 and d_eps.is_contiguous()
        assert d_eps_pocket is None or (d_eps_pocket.is_cuda and d_eps_pocket.dtype == torch.float32 and d_eps_pocket.is_contiguous())
        assert grad.is_cuda and grad.dtype == torch.float32 and grad.is_contiguous() and grad.numel() == self.param_count()
        self._check(self.lib.cmdgen_train_backward(self.h, _ptr(d_eps), _ptr(d_eps_pocket), _ptr(grad), self._stream()),
                    'cmdgen_train_backward')

    def train_backward_stages(self, d_eps, grad, first_stage: int, last_stage: int, d_eps_pocket=None):
        """Stages first..last of the backward pass (0 readout, k = block L-k, L+1 embedding / encoders)."""
        self._check(self.lib.cmdgen_train_backward_stages(self.h, _ptr(d_eps), _ptr(d_eps_pocket), _ptr(grad), int(first_stage),
                                                          int(last_stage), self._stream()), 'cmdgen_train_backward_stages')

    def train_set_precision(self, bf16_gemm: bool):
        self._check(self.lib.cmdgen_train_set_precision(self.h, int(bool(bf16_gemm))), 'cmdgen_train_set_precision')

    TT_COLS, TS_COLS = 12, 12        # include/cmdgen_hip.h: CMDGEN_TT_COLS / CMDGEN_TS_COLS

    def train_noise(self, phar_x, phar_one_hot, pocket_x, pocket_one_hot, tab, eps):
        """-> (z_t, xh_pocket, kl_sums): the fused noising of the conditional training step (cmdgen_train_noise)."""
        import torch
        P, R = self.cfg['phar_nf'], self.cfg['residue_nf']
        for t, shape in ((phar_x, (self.n_phar, 3)), (phar_one_hot, (self.n_phar, P)), (pocket_x, (self.n_pocket, 3)),
                         (pocket_one_hot, (self.n_pocket, R)), (tab, (self.TT_COLS, self.batch)), (eps, (self.n_phar, 3 + P))):
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == shape, (tuple(t.shape), shape)
        z_t = torch.empty((self.n_phar, 3 + P), dtype=torch.float32, device=eps.device)
        xh_pocket = torch.empty((self.n_pocket, 3 + R), dtype=torch.float32, device=eps.device)
        kl = torch.empty((self.batch, 2), dtype=torch.float32, device=eps.device)
        self._check(self.lib.cmdgen_train_noise(self.h, _ptr(phar_x), _ptr(phar_one_hot), _ptr(pocket_x), _ptr(pocket_one_hot),
                                                _ptr(tab), _ptr(eps), _ptr(z_t), _ptr(xh_pocket), _ptr(kl), self._stream()),
                    'cmdgen_train_noise')
        return z_t, xh_pocket, kl

    def train_loss(self, l2: bool, T: float, net_out, eps, z_t, phar_one_hot, tab, kl_sums):
        """-> (terms [B, TS_COLS], means [TS_COLS], d_eps): per-sample loss terms, their batch means and d loss / d net_out."""
        import torch
        assert net_out.is_cuda and net_out.is_contiguous() and net_out.shape == eps.shape == z_t.shape
        terms = torch.zeros((self.batch, self.TS_COLS), dtype=torch.float32, device=eps.device)
        means = torch.empty(self.TS_COLS, dtype=torch.float32, device=eps.device)
        d_eps = torch.empty_like(net_out)
        self._check(self.lib.cmdgen_train_loss(self.h, int(bool(l2)), float(T), _ptr(net_out), _ptr(eps), _ptr(z_t), _ptr(phar_one_hot),
                                               _ptr(tab), _ptr(kl_sums), _ptr(terms), _ptr(d_eps), _ptr(means), self._stream()),
                    'cmdgen_train_loss')
        return terms, means, d_eps

    def train_noise_joint(self, phar_x, phar_one_hot, pocket_x, pocket_one_hot, tab, draw_phar, draw_pocket):
        """-> (z_phar, z_pocket, eps_phar, eps_pocket, kl_sums): the fused noising of the joint model's training step."""
        import torch
        P, R = self.cfg['phar_nf'], self.cfg['residue_nf']
        for t, shape in ((phar_x, (self.n_phar, 3)), (phar_one_hot, (self.n_phar, P)), (pocket_x, (self.n_pocket, 3)),
                         (pocket_one_hot, (self.n_pocket, R)), (tab, (self.TT_COLS, self.batch)), (draw_phar, (self.n_phar, 3 + P)),
                         (draw_pocket, (self.n_pocket, 3 + R))):
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == shape, (tuple(t.shape), shape)
        z_l, z_q, e_l, e_q = (torch.empty_like(draw_phar), torch.empty_like(draw_pocket), torch.empty_like(draw_phar),
                              torch.empty_like(draw_pocket))
        kl = torch.empty((self.batch, 2), dtype=torch.float32, device=tab.device)
        self._check(self.lib.cmdgen_train_noise_joint(self.h, _ptr(phar_x), _ptr(phar_one_hot), _ptr(pocket_x), _ptr(pocket_one_hot), _ptr(tab),
                                                      _ptr(draw_phar), _ptr(draw_pocket), _ptr(z_l), _ptr(z_q), _ptr(e_l), _ptr(e_q), _ptr(kl),
                                                      self._stream()), 'cmdgen_train_noise_joint')
        return z_l, z_q, e_l, e_q, kl

    def train_loss_joint(self, l2: bool, T: float, net_phar, net_pocket, eps_phar, eps_pocket, z_phar, z_pocket, phar_one_hot,
                         pocket_one_hot, tab, kl_sums):
        """-> (terms [B, TS_COLS], means [TS_COLS], d_eps_phar, d_eps_pocket) of the joint model's training loss."""
        import torch
        for a, b in ((net_phar, eps_phar), (net_pocket, eps_pocket)):
            assert a.is_cuda and a.is_contiguous() and a.dtype == torch.float32 and a.shape == b.shape
        terms = torch.zeros((self.batch, self.TS_COLS), dtype=torch.float32, device=tab.device)
        means = torch.empty(self.TS_COLS, dtype=torch.float32, device=tab.device)
        d_l, d_q = torch.empty_like(net_phar), torch.empty_like(net_pocket)
        self._check(self.lib.cmdgen_train_loss_joint(self.h, int(bool(l2)), float(T), _ptr(net_phar), _ptr(net_pocket), _ptr(eps_phar),
                                                     _ptr(eps_pocket), _ptr(z_phar), _ptr(z_pocket), _ptr(phar_one_hot), _ptr(pocket_one_hot),
                                                     _ptr(tab), _ptr(kl_sums), _ptr(terms), _ptr(d_l), _ptr(d_q), _ptr(means), self._stream()),
                    'cmdgen_train_loss_joint')
        return terms, means, d_l, d_q

    def grad_sqnorm(self, grad) -> float:
        out = C.c_float(0)
        self._check(self.lib.cmdgen_grad_sqnorm(self.h, _ptr(grad), grad.numel(), C.byref(out), self._stream()), 'cmdgen_grad_sqnorm')
        return out.value

    def adamw_step(self, theta, grad, exp_avg, exp_avg_sq, max_exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8,
                   weight_decay=1e-12, clip_coef=1.0):
        self._check(self.lib.cmdgen_adamw_step(self.h, _ptr(theta), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq),
                                               _ptr(max_exp_avg_sq), theta.numel(), int(step), float(lr), float(betas[0]),
                                               float(betas[1]), float(eps), float(weight_decay), float(clip_coef),
                                               self._stream()), 'cmdgen_adamw_step')

    def adamw_step_clipped(self, theta, grad, exp_avg, exp_avg_sq, max_exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8,
                           weight_decay=1e-12, max_grad_norm=0.0, defer: bool = False):
        """Norm + clipping (coefficient formed on the device; max_grad_norm <= 0: none) + AdamW; returns the gradient norm,
        or None with defer=True (collect it with last_grad_norm(); nothing is waited for here)."""
        out = C.c_float(0)
        self._check(self.lib.cmdgen_adamw_step_clipped(self.h, _ptr(theta), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq),
                                                       _ptr(max_exp_avg_sq), theta.numel(), int(step), float(lr), float(betas[0]),
                                                       float(betas[1]), float(eps), float(weight_decay), float(max_grad_norm),
                                                       None if defer else C.byref(out), self._stream()), 'cmdgen_adamw_step_clipped')
        return None if defer else out.value

    def last_grad_norm(self) -> float:
        out = C.c_float(0)
        self._check(self.lib.cmdgen_last_grad_norm(self.h, C.byref(out)), 'cmdgen_last_grad_norm')
        return out.value

    def debug_wgrad(self, dY, X, dW, db=None, mode=0):
        """dW += dY^T X, db += column sums of dY through the weight-gradient launch (test aid); mode 0 fp32, 1 bf16 operands, 3 split."""
        K, M = dY.shape
        N = X.shape[1]
        assert X.shape[0] == K and tuple(dW.shape) == (M, N) and dY.is_contiguous() and X.is_contiguous() and dW.is_contiguous()
        self._check(self.lib.cmdgen_debug_wgrad(self.h, K, M, N, _ptr(dY), _ptr(X), _ptr(dW), _ptr(db), int(mode), self._stream()),
                    'cmdgen_debug_wgrad')
        return dW

    def debug_dgrad(self, A0, W0, A1=None, W1=None, Y=None, accumulate=False, div=1.0, pre=None, pieces=3, tile_rows=0):
        """Y (+)= (A0 W0 + A1 W1) / div * SiLU'(pre) through k_dgrad_split (test aid); W0 / W1 views of one [2, 256, 256] tensor."""
        import torch
        M = A0.shape[0]
        out = torch.zeros((M, 256), dtype=torch.float32, device=A0.device) if Y is None else Y
        self._check(self.lib.cmdgen_debug_dgrad(self.h, M, _ptr(A0), _ptr(W0), _ptr(A1), _ptr(W1), _ptr(out), int(accumulate), float(div),
                                                _ptr(pre), int(pieces), int(tile_rows), self._stream()), 'cmdgen_debug_dgrad')
        return out

    def debug_sgemm(self, A, B, ta=False, tb=True, bias=None, C_out=None, accumulate=False, split_k=1, bf16=False):
        import torch
        M = A.shape[1] if ta else A.shape[0]
        K = A.shape[0] if ta else A.shape[1]
        N = B.shape[0] if tb else B.shape[1]
        out = torch.zeros((M, N), dtype=torch.float32, device=A.device) if C_out is None else C_out
        self._check(self.lib.cmdgen_debug_sgemm(self.h, int(ta), int(tb), M, N, K, _ptr(A), A.stride(0), _ptr(B), B.stride(0),
                                                _ptr(out), out.stride(0), _ptr(bias), int(accumulate) | (2 if bf16 else 0), int(split_k),
                                                self._stream()), 'cmdgen_debug_sgemm')
        return out

    def chain_status(self):
        """Deferred checks of the last chain.  `nan_resets` counts the NaN reset steps since the previous call (the library's counter is
        cumulative: `nan_resets_total`)."""
        a, b, n = C.c_float(0), C.c_float(0), C.c_int64(0)
        self._check(self.lib.cmdgen_chain_status(self.h, C.byref(a), C.byref(b), C.byref(n), self._stream()),
                    'cmdgen_chain_status')
        seen = getattr(self, '_nan_seen', 0)
        self._nan_seen = n.value
        return {'max_rel_com_error': a.value, 'max_cog': b.value, 'nan_resets': max(n.value - seen, 0), 'nan_resets_total': n.value}

    # ---- the half engine's range (two fp16 pieces per operand: an activation beyond 65504 becomes Inf -> NaN -> a reset step the fp32 reference
    # does not take).  A NaN reset on a half-engine handle is therefore never accepted as it stands: the call is repeated on the three-piece bf16
    # split engine (fp32's exponent range) with the same inputs and draws.  If that run is clean, its result is returned (and a warning names the
    # cause); if it resets too, the NaN is the model's own - the reference resets there as well (dynamics.py:129-131) - and that run is returned.
    def half_engine_active(self) -> bool:
        return bool(self.query('half_engine'))

    def run_range_guarded(self, run, status):
        """run() queues the work and returns its outputs; status() -> dict with 'nan_resets' of that run.  -> (outputs, status dict)."""
        out = run()
        st = status()
        if not st['nan_resets'] or not self.half_engine_active():
            return out, st
        import warnings
        prev = self.get_option('half_engine')
        self.set_option('half_engine', 0)
        try:
            out2 = run()
            st2 = status()
        finally:
            self.set_option('half_engine', prev)
        st2['half_engine_fallback'] = True
        if not st2['nan_resets']:
            warnings.warn('an activation left the half matrix engine\'s range (|a| > 65504): this call was repeated on the three-piece bf16 engine '
                          '(set option half_engine=0 on this handle to run there from the start)', RuntimeWarning, stacklevel=3)
        return out2, st2

    def nan_resets_total(self) -> int:
        return self.counters()['nan_resets']

    # ---- measurement
    def counters(self) -> Dict[str, int]:
        c = Counters()
        self._check(self.lib.cmdgen_get_counters(self.h, C.byref(c), self._stream()), 'cmdgen_get_counters')
        return {k: int(getattr(c, k)) for k in ('evaluations', 'edges', 'edges_phar', 'nodes', 'nan_resets', 'edges_skipped', 'node_rows_skipped')}

    def reset_counters(self):
        self._check(self.lib.cmdgen_reset_counters(self.h, self._stream()), 'cmdgen_reset_counters')
        self._nan_seen = 0

    def profile_evaluation(self, xh_phar, xh_pocket, t):
        import torch
        t = t.reshape(-1).to(torch.float32).contiguous()
        eps = torch.empty_like(xh_phar)
        kt = KernelTimes()
        self._check(self.lib.cmdgen_profile_evaluation(self.h, _ptr(xh_phar), _ptr(xh_pocket), _ptr(t), _ptr(eps),
                                                       C.byref(kt), self._stream()), 'cmdgen_profile_evaluation')
        return {n: getattr(kt, n) for n, _ in KernelTimes._fields_}

    def set_gemm_mode(self, split_bf16: bool) -> None:
        """Matrix engine of tiles of >= 32 rows (sampler and training step): True = split-bf16 (fp32-accurate, default), False = fp32 MFMA."""
        self._check(self.lib.cmdgen_set_gemm_mode(self.h, int(bool(split_bf16))), 'cmdgen_set_gemm_mode')

    def set_option(self, key: str, value: Optional[int]) -> None:
        """An explicit launch choice of this handle (include/cmdgen_hip.h, cmdgen_set_option); value None = back to the library's own choice."""
        self._check(self.lib.cmdgen_set_option(self.h, key.encode(), int(value or 0), int(value is None)), 'cmdgen_set_option')

    def get_option(self, key: str) -> Optional[int]:
        v, isset = C.c_int64(0), C.c_int32(0)
        self._check(self.lib.cmdgen_get_option(self.h, key.encode(), C.byref(v), C.byref(isset)), 'cmdgen_get_option')
        return v.value if isset.value else None

    def debug_stamps(self, reset: bool = True):
        """Diagnostic builds (-DCMDGEN_STAMPS): the 64 summed in-kernel cycle counters (zero in production builds)."""
        out = (C.c_uint64 * 64)()
        self._check(self.lib.cmdgen_debug_stamps(self.h, out, int(reset)), 'cmdgen_debug_stamps')
        return list(out)

    def query(self, key: str) -> int:
        v = C.c_int64(0)
        self._check(self.lib.cmdgen_query(self.h, key.encode(), C.byref(v)), 'cmdgen_query')
        return v.value

    def time_evaluation(self, xh_phar, xh_pocket, t, graph_len: int = 10, replays: int = 10) -> float:
        """ms per evaluation, graph-replayed and timed with HIP events on the launch stream."""
        import torch
        t = t.reshape(-1).to(torch.float32).contiguous()
        eps = torch.empty_like(xh_phar)
        ms = C.c_float(0)
        self._check(self.lib.cmdgen_time_evaluation(self.h, _ptr(xh_phar), _ptr(xh_pocket), _ptr(t), _ptr(eps), int(graph_len),
                                                    int(replays), C.byref(ms), self._stream()), 'cmdgen_time_evaluation')
        return ms.value

    def time_edge_kernel(self, layer: int, reps: int) -> float:
        ms = C.c_float(0)
        self._check(self.lib.cmdgen_time_edge_kernel(self.h, layer, reps, C.byref(ms), self._stream()),
                    'cmdgen_time_edge_kernel')
        return ms.value

    def set_kernel_profiling(self, on: bool):
        self._check(self.lib.cmdgen_set_kernel_profiling(self.h, int(bool(on))), 'cmdgen_set_kernel_profiling')

    def kernel_profile(self):
        """-> {'edge_msg'|'node'|'edge_coord': (summed ms, launches)} since the last call (eager chains only)."""
        ms = (C.c_float * 3)()
        n = (C.c_int64 * 3)()
        self._check(self.lib.cmdgen_get_kernel_profile(self.h, ms, n, self._stream()), 'cmdgen_get_kernel_profile')
        return {k: (ms[i], n[i]) for i, k in enumerate(('edge_msg', 'node', 'edge_coord'))}

    def debug_noise(self, seed: int, pocket_id: int, draw: int, n_nodes: int, width: int = 11):
        import torch
        out = torch.empty((n_nodes, width), dtype=torch.float32, device=f'cuda:{self.device_index}')
        self._check(self.lib.cmdgen_debug_noise(self.h, C.c_uint64(seed), C.c_int64(pocket_id), draw, n_nodes, width,
                                                _ptr(out), self._stream()), 'cmdgen_debug_noise')
        return out
