"""EGNNDynamics: the denoiser eps = f(z_phar, pocket, t) (counterpart of dynamics.py:9-147).

Same constructor, parameter names and ``forward`` signature as the reference; the forward
pass itself is ONE call into libcmdgen_hip.so (radius graph, encoders, L fused EGNN blocks,
decoders, NaN guard) - see csrc/kernels_egnn.hip.  No CPU path exists.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from .egnn_new import EGNN
from .. import hip_backend
from ..utils import sizes_from_mask


class EGNNDynamics(nn.Module):
    def __init__(self, phar_nf, residue_nf, n_dims, joint_nf=16, hidden_nf=64, device='cpu',
                 act_fn=torch.nn.SiLU(), n_layers=4, attention=False, condition_time=True, tanh=False,
                 mode='egnn_dynamics', norm_constant=0, inv_sublayers=2, sin_embedding=False,
                 normalization_factor=100, aggregation_method='sum', update_pocket_coords=True,
                 edge_cutoff=None):
        super().__init__()
        if mode != 'egnn_dynamics':
            raise NotImplementedError("mode 'gnn_dynamics' is not used by the shipped configs and not built")
        self.mode, self.edge_cutoff = mode, edge_cutoff
        self.phar_encoder = nn.Sequential(nn.Linear(phar_nf, 2 * phar_nf), act_fn, nn.Linear(2 * phar_nf, joint_nf))
        self.phar_decoder = nn.Sequential(nn.Linear(joint_nf, 2 * phar_nf), act_fn, nn.Linear(2 * phar_nf, phar_nf))
        self.residue_encoder = nn.Sequential(nn.Linear(residue_nf, 2 * residue_nf), act_fn,
                                             nn.Linear(2 * residue_nf, joint_nf))
        self.residue_decoder = nn.Sequential(nn.Linear(joint_nf, 2 * residue_nf), act_fn,
                                             nn.Linear(2 * residue_nf, residue_nf))
        dynamics_node_nf = joint_nf + 1 if condition_time else joint_nf
        self.egnn = EGNN(in_node_nf=dynamics_node_nf, in_edge_nf=1, hidden_nf=hidden_nf, act_fn=act_fn,
                         n_layers=n_layers, attention=attention, tanh=tanh, norm_constant=norm_constant,
                         inv_sublayers=inv_sublayers, sin_embedding=sin_embedding,
                         normalization_factor=normalization_factor, aggregation_method=aggregation_method)
        self.node_nf = dynamics_node_nf
        self.update_pocket_coords = update_pocket_coords
        self.device = device
        self.n_dims = n_dims
        self.condition_time = condition_time
        self._cfg = dict(phar_nf=phar_nf, residue_nf=residue_nf, joint_nf=joint_nf, hidden_nf=hidden_nf,
                         n_layers=n_layers, inv_sublayers=inv_sublayers, attention=attention, tanh=tanh,
                         condition_time=condition_time, edge_cutoff=edge_cutoff, norm_constant=norm_constant,
                         normalization_factor=normalization_factor, aggregation_method=aggregation_method,
                         sin_embedding=sin_embedding, coords_range=self.egnn.coords_range,
                         update_pocket_coords=bool(update_pocket_coords), timesteps=1, norm_values=(1.0, 1.0), norm_biases=(None, 0.0))
        self._gamma = np.zeros(2, dtype=np.float32)
        self._handle = None
        self._weights_sig = None
        self._mask_cache = None

    # -- wiring from the diffusion module (schedule table and normalisation live in the same handle)
    def attach_diffusion(self, timesteps, gamma_table, norm_values, norm_biases, no_com_projection=False):
        self._cfg.update(timesteps=int(timesteps), norm_values=tuple(norm_values), norm_biases=tuple(norm_biases),
                         no_com_projection=bool(no_com_projection))
        self._gamma = np.asarray(gamma_table, dtype=np.float32)
        self._handle, self._weights_sig = None, None

    def _signature(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def hip_handle(self) -> "hip_backend.Handle":
        """The cmdgen handle for this module's device, with the current weights uploaded."""
        p0 = next(self.parameters())
        if p0.device.type != 'cuda':
            raise hip_backend.CmdgenError(
                'EGNNDynamics runs on MI355X only: move the module to cuda (there is no CPU fallback)')
        idx = p0.device.index if p0.device.index is not None else torch.cuda.current_device()
        if self._handle is None or self._handle.device_index != idx:
            self._handle = hip_backend.Handle(self._cfg, idx)
            self._weights_sig = None
        sig = self._signature()
        if sig != self._weights_sig:
            state = {'ddpm.dynamics.' + k: v for k, v in self.state_dict().items()}
            state['ddpm.gamma.gamma'] = self._gamma
            self._handle.load_state_dict(state)
            self._weights_sig = sig
        return self._handle

    def _layout_from_masks(self, mask_phars, mask_residues, batch):
        key = (mask_phars.data_ptr(), mask_phars._version, len(mask_phars),
               mask_residues.data_ptr(), mask_residues._version, len(mask_residues), batch)
        if self._mask_cache is None or self._mask_cache[0] != key:
            self._mask_cache = (key, sizes_from_mask(mask_phars, batch), sizes_from_mask(mask_residues, batch))
        return self._mask_cache[1], self._mask_cache[2]

    def forward(self, xh_phars, xh_residues, t, mask_phars, mask_residues):
        """-> (eps_phar [Nl, 3+phar_nf], eps_pocket [Np, 3+residue_nf]); dynamics.py:75-139."""
        h = self.hip_handle()
        batch = int(t.numel()) if t.numel() > 1 else int(max(int(mask_phars.max()), int(mask_residues.max())) + 1)
        nph, npk = self._layout_from_masks(mask_phars, mask_residues, batch)
        h.set_layout(nph, npk)
        xp = xh_phars.detach().to(torch.float32).contiguous()
        xr = xh_residues.detach().to(torch.float32).contiguous()
        # one evaluation: the NaN guard's counter tells whether a reset happened (the reference syncs here too: `torch.any(torch.isnan(vel))`,
        # dynamics.py:129); on the half matrix engine such a reset is re-run on the bf16 split engine before it is believed
        if not h.half_engine_active():
            return h.dynamics_forward(xp, xr, t.detach(), want_pocket=True)
        seen = [h.nan_resets_total()]

        def status():
            now = h.nan_resets_total()
            d, seen[0] = now - seen[0], now
            return {'nan_resets': max(d, 0)}
        out, _ = h.run_range_guarded(lambda: h.dynamics_forward(xp, xr, t.detach(), want_pocket=True), status)
        return out

    def get_edges(self, batch_mask=None, x=None):
        """Radius graph as [2, E] int64: pairs (i, j), self loops included, with batch_mask[i] == batch_mask[j] and
        ||x_i - x_j|| <= edge_cutoff, sorted by (i, j) like torch.where on the adjacency (dynamics.py:141-147).

        With arguments: the graph of exactly what is given (any mask, e.g. the phar-rows-first concatenation
        forward() uses); the distance tests and the compaction run in the HIP radius-graph kernels, torch only
        sorts indices.  Without arguments: the graph the last forward() built (kept for inspection)."""
        if batch_mask is None and x is None:
            if self._handle is None:
                raise RuntimeError('get_edges() without arguments returns the graph of the last forward(); call forward first')
            return torch.from_numpy(self._handle.get_edges().astype(np.int64))
        h = self.hip_handle()
        dev = next(self.parameters()).device
        mask = batch_mask.detach().to(dev, torch.int64).reshape(-1)
        xx = x.detach().to(dev, torch.float32)
        n = mask.numel()
        if n == 0:
            return torch.zeros((2, 0), dtype=torch.int64, device=dev)
        order = torch.sort(mask, stable=True).indices                 # samples back to back, original order inside
        counts = torch.bincount(mask[order]).cpu().numpy()
        row, col = h.radius_graph(xx[order].contiguous(), counts[counts > 0])
        row, col = order[row.long()], order[col.long()]               # back to the caller's numbering
        key = torch.sort(row * n + col).indices                       # torch.where order: by (row, col)
        return torch.stack([row[key], col[key]])
