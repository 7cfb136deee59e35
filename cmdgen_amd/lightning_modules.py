"""PharPocketDDPM: the module callers import (counterpart of lightning_modules.py:30-568).

Same constructor arguments, checkpoint format (Lightning 1.8.5 ``.ckpt``: ``state_dict`` with
``ddpm.`` keys + ``hyper_parameters``), ``generate_phars`` and sampling entry points - without
a pytorch_lightning / BioPython / RDKit dependency.  Everything numerical on the sampling
path runs in libcmdgen_hip.so.  Training plumbing (Lightning hooks, W&B, dataloaders) is out
of scope.
"""
from __future__ import annotations

import argparse
import math
from argparse import Namespace
from typing import Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .constants import dataset_params, FLOAT_TYPE, INT_TYPE
from .equivariant_diffusion.dynamics import EGNNDynamics
from .equivariant_diffusion.en_diffusion import EnVariationalDiffusion
from .equivariant_diffusion.conditional_model import ConditionalDDPM, SimpleConditionalDDPM
from . import utils


def _scatter_mean(src, index, n):
    tot = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device).index_add_(0, index, src)
    cnt = torch.zeros(n, dtype=src.dtype, device=src.device).index_add_(
        0, index, torch.ones(len(index), dtype=src.dtype, device=src.device)).clamp(min=1)
    return tot / cnt.view((-1,) + (1,) * (src.dim() - 1))


class PharPocketDDPM(nn.Module):
    def __init__(self, outdir, dataset, datadir, batch_size, lr, egnn_params: Namespace, diffusion_params,
                 num_workers, augment_noise, augment_rotation, clip_grad, eval_epochs, eval_params, mode,
                 node_histogram, pocket_representation='CA'):
        super().__init__()
        self.hparams = dict(outdir=outdir, dataset=dataset, datadir=datadir, batch_size=batch_size, lr=lr,
                            egnn_params=egnn_params, diffusion_params=diffusion_params, num_workers=num_workers,
                            augment_noise=augment_noise, augment_rotation=augment_rotation, clip_grad=clip_grad,
                            eval_epochs=eval_epochs, eval_params=eval_params, mode=mode,
                            node_histogram=node_histogram, pocket_representation=pocket_representation)
        ddpm_models = {'joint': EnVariationalDiffusion, 'pocket_conditioning': ConditionalDDPM,
                       'pocket_conditioning_simple': SimpleConditionalDDPM}
        assert mode in ddpm_models
        self.mode = mode
        assert pocket_representation in {'CA', 'full-atom'}
        self.pocket_representation = pocket_representation
        self.dataset_name, self.datadir, self.outdir = dataset, datadir, outdir
        self.batch_size = batch_size
        ep = vars(eval_params) if isinstance(eval_params, Namespace) else dict(eval_params or {})
        self.eval_batch_size = ep.get('eval_batch_size', batch_size)
        self.lr = lr
        self.loss_type = diffusion_params.diffusion_loss_type
        self.eval_epochs, self.eval_params = eval_epochs, eval_params
        self.num_workers, self.augment_noise, self.augment_rotation = num_workers, augment_noise, augment_rotation
        self.dataset_info = dataset_params[dataset]
        self.T = diffusion_params.diffusion_steps
        self.clip_grad = clip_grad
        if clip_grad:
            self.gradnorm_queue = utils.Queue()
            self.gradnorm_queue.add(3000)       # large value that will be flushed (lightning_modules.py:78-80)
        self.phar_type_encoder = self.dataset_info['phar_encoder']
        self.phar_type_decoder = self.dataset_info['phar_decoder']
        ca = self.pocket_representation == 'CA'
        self.pocket_type_encoder = self.dataset_info['aa_encoder' if ca else 'atom_encoder']
        self.pocket_type_decoder = self.dataset_info['aa_decoder' if ca else 'atom_decoder']
        self.phar_nf = len(self.phar_type_decoder)
        self.aa_nf = len(self.pocket_type_decoder)
        self.x_dims = 3
        net_dynamics = EGNNDynamics(
            phar_nf=self.phar_nf, residue_nf=self.aa_nf, n_dims=self.x_dims, joint_nf=egnn_params.joint_nf,
            device='cpu', hidden_nf=egnn_params.hidden_nf, act_fn=torch.nn.SiLU(), n_layers=egnn_params.n_layers,
            attention=egnn_params.attention, tanh=egnn_params.tanh, norm_constant=egnn_params.norm_constant,
            inv_sublayers=egnn_params.inv_sublayers, sin_embedding=egnn_params.sin_embedding,
            normalization_factor=egnn_params.normalization_factor,
            aggregation_method=egnn_params.aggregation_method,
            edge_cutoff=egnn_params.__dict__.get('edge_cutoff'), update_pocket_coords=(self.mode == 'joint'))
        self.ddpm = ddpm_models[self.mode](
            dynamics=net_dynamics, phar_nf=self.phar_nf, residue_nf=self.aa_nf, n_dims=self.x_dims,
            timesteps=diffusion_params.diffusion_steps,
            noise_schedule=diffusion_params.diffusion_noise_schedule,
            noise_precision=diffusion_params.diffusion_noise_precision,
            loss_type=diffusion_params.diffusion_loss_type,
            norm_values=diffusion_params.normalize_factors, size_histogram=node_histogram)

    # ------------------------------------------------------------------ Lightning-free plumbing
    @property
    def device(self):
        return next(self.parameters()).device

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location=None, trust_checkpoint: bool = False, **overrides):
        """Read a Lightning-format checkpoint (generate_phars.py:32, test.py:73): a pickled dict with
        'state_dict' and 'hyper_parameters' (every constructor argument, incl. argparse.Namespace)."""
        if trust_checkpoint:
            ckpt = torch.load(checkpoint_path, map_location=map_location, weights_only=False)
        else:
            import numpy.core.multiarray as _ncm
            import pathlib
            # the reference passes outdir as a pathlib.Path (train.py:64-66), so real checkpoints carry one
            safe = [argparse.Namespace, np.ndarray, np.dtype, _ncm._reconstruct, _ncm.scalar,
                    pathlib.PosixPath, pathlib.PurePosixPath, pathlib.Path, pathlib.PurePath]
            try:
                safe += [type(np.dtype(np.float64)), type(np.dtype(np.int64)), type(np.dtype(np.float32))]
            except Exception:
                pass
            with torch.serialization.safe_globals(safe):
                ckpt = torch.load(checkpoint_path, map_location=map_location, weights_only=True)
        hp = dict(ckpt['hyper_parameters'])
        hp.update(overrides)
        for k in ('egnn_params', 'diffusion_params', 'eval_params'):
            if isinstance(hp.get(k), dict):
                hp[k] = Namespace(**hp[k])
        model = cls(**hp)
        model.load_state_dict(ckpt['state_dict'], strict=True)
        return model

    def save_checkpoint(self, path):
        torch.save({'state_dict': self.state_dict(), 'hyper_parameters': dict(self.hparams)}, path)

    def get_phar_and_pocket(self, data):
        phar = {'x': data['phar_coords'].to(self.device, FLOAT_TYPE),
                'one_hot': data['phar_one_hot'].to(self.device, FLOAT_TYPE),
                'size': data['num_phar_atoms'].to(self.device, INT_TYPE),
                'mask': data['phar_mask'].to(self.device, INT_TYPE)}
        pocket = {'x': data['pocket_c_alpha'].to(self.device, FLOAT_TYPE),
                  'one_hot': data['pocket_one_hot'].to(self.device, FLOAT_TYPE),
                  'size': data['num_pocket_nodes'].to(self.device, INT_TYPE),
                  'mask': data['pocket_mask'].to(self.device, INT_TYPE)}
        return phar, pocket

    def forward(self, data, t_int=None, eps=None, _net=None):
        """-> (nll [B], info) as lightning_modules.py:188-239.  Loss VALUES (evaluation / monitoring) - no autograd
        graph is built; the optimizer is driven by training.HipTrainer, which runs the HIP library's own backward
        pass (cmdgen_train_forward / cmdgen_train_backward) with the analytic gradient of this loss."""
        phar, pocket = self.get_phar_and_pocket(data)
        delta_log_px, error_t_phar, error_t_pocket, SNR_weight, loss_0_x_phar, loss_0_x_pocket, loss_0_h, \
            neg_log_const_0, kl_prior, log_pN, t_int_, xh_phar_hat, info = \
            self.ddpm(phar, pocket, return_info=True, t_int=t_int, eps=eps, **({'_net': _net} if _net is not None else {}))
        dev = error_t_phar.device
        error_t_pocket, loss_0_x_pocket = error_t_pocket.to(dev), loss_0_x_pocket.to(dev)
        if self.loss_type == 'l2' and self.training:
            error_t_phar = error_t_phar / ((self.x_dims + self.ddpm.phar_nf) * phar['size'])
            error_t_pocket = error_t_pocket / ((self.x_dims + self.ddpm.residue_nf) * pocket['size'])
            loss_t = 0.5 * (error_t_phar + error_t_pocket)
            loss_0 = loss_0_x_phar / (self.x_dims * phar['size']) + loss_0_x_pocket / (self.x_dims * pocket['size']) \
                + loss_0_h
        else:
            loss_t = -self.T * 0.5 * SNR_weight * (error_t_phar + error_t_pocket)
            loss_0 = loss_0_x_phar + loss_0_x_pocket + loss_0_h + neg_log_const_0
        nll = loss_t + loss_0 + kl_prior
        if not (self.loss_type == 'l2' and self.training):
            nll = nll - delta_log_px - log_pN        # normalisation on x; conditional -> joint nll
        info['error_t_phar'] = error_t_phar.mean(0)
        info['error_t_pocket'] = error_t_pocket.mean(0)
        info['SNR_weight'] = SNR_weight.mean(0)
        info['loss_0'] = loss_0.mean(0)
        info['kl_prior'] = kl_prior.mean(0)
        info['delta_log_px'] = delta_log_px.mean(0)
        info['neg_log_const_0'] = neg_log_const_0.mean(0)
        info['log_pN'] = log_pN.mean(0)
        return nll, info

    def setup(self, stage=None):
        """Datasets from <datadir>/{train,val,test}.npz (lightning_modules.py:145-155)."""
        from pathlib import Path
        from .dataset import ProcessedLigandPharPocketDataset
        if stage == 'fit':
            self.train_dataset = ProcessedLigandPharPocketDataset(Path(self.datadir, 'train.npz'))
            self.val_dataset = ProcessedLigandPharPocketDataset(Path(self.datadir, 'val.npz'))
        elif stage == 'test':
            self.test_dataset = ProcessedLigandPharPocketDataset(Path(self.datadir, 'test.npz'))
        else:
            raise NotImplementedError

    @torch.no_grad()
    def sample_given_pocket_dataset(self, n_samples, dataset, batch_size=None, timesteps=None, **kw):
        """Sampling loop of sample_and_analyze_given_pocket (lightning_modules.py:337-373): cycles through the
        dataset in batches, draws the phar-node counts from the size prior and samples; returns per-sample
        (coords, types, reference phar coords).  The RDKit / KL analysis that follows in the reference
        (analysis/metrics.py) is out of scope."""
        batch_size = self.batch_size if batch_size is None else batch_size
        batch_size = min(batch_size, n_samples)
        phars = []
        for i in range(math.ceil(n_samples / batch_size)):
            n_b = min(batch_size, n_samples - len(phars))
            batch = dataset.collate_fn([dataset[(i * batch_size + j) % len(dataset)] for j in range(n_b)])
            phar, pocket = self.get_phar_and_pocket(batch)
            num_nodes_phar = self.ddpm.size_distribution.sample_conditional(n1=None, n2=pocket['size'])
            xh_phar, xh_pocket, phar_mask, _ = self.ddpm.sample_given_pocket(pocket, num_nodes_phar,
                                                                            timesteps=timesteps, **kw)
            x = xh_phar[:, :self.x_dims].detach().cpu()
            t = xh_phar[:, self.x_dims:].argmax(1).detach().cpu()
            pm = phar_mask.cpu()
            phars.extend(zip(utils.batch_to_list(x, pm), utils.batch_to_list(t, pm),
                             utils.batch_to_list(phar['x'].cpu(), phar['mask'].cpu())))
        return phars

    # ------------------------------------------------------------------ validation sampling
    @staticmethod
    def _type_kl(hist_dict, mapping, sample_types):
        """KL(p || q) of the dataset's type histogram p against the sampled types q with the reference's smoothing
        (analysis/metrics.py:12-34, CategoricalDistribution.kl_divergence; -1 when there is no histogram)."""
        if hist_dict is None:
            return -1
        p = np.zeros(len(mapping))
        for k, v in hist_dict.items():
            p[mapping[k]] = v
        p = p / p.sum()
        q = np.bincount(np.asarray(sample_types, dtype=np.int64), minlength=len(mapping)).astype(np.float64)
        q = q / q.sum()
        with np.errstate(divide='ignore', invalid='ignore'):
            return float(-np.sum(p * np.log(q / p + 1e-10)))

    @torch.no_grad()
    def sample_and_analyze_given_pocket(self, n_samples, dataset=None, batch_size=None, timesteps=None, **kw):
        """Sample n_samples pharmacophores for pockets of `dataset` and compare type statistics with the training
        data (lightning_modules.py:337-382 + analyze_sample :307-334): KL divergence of the sampled phar types and of
        the pocket types the sampler returns.  The commented-out stability / RDKit parts of the reference are not
        reproduced."""
        batch_size = self.batch_size if batch_size is None else batch_size
        batch_size = min(batch_size, n_samples)
        phar_types, aa_types, done = [], [], 0
        for i in range(math.ceil(n_samples / batch_size)):
            n_b = min(batch_size, n_samples - done)
            batch = dataset.collate_fn([dataset[(i * batch_size + j) % len(dataset)] for j in range(n_b)])
            phar, pocket = self.get_phar_and_pocket(batch)
            num_nodes_phar = self.ddpm.size_distribution.sample_conditional(n1=None, n2=pocket['size'])
            if type(self.ddpm) == EnVariationalDiffusion:
                xh_phar, xh_pocket, _, _ = self.ddpm.sample(n_b, num_nodes_phar, pocket['size'], timesteps=timesteps,
                                                            device=self.device, **kw)
            else:
                xh_phar, xh_pocket, _, _ = self.ddpm.sample_given_pocket(pocket, num_nodes_phar, timesteps=timesteps, **kw)
            phar_types.extend(xh_phar[:, self.x_dims:].argmax(1).detach().cpu().tolist())
            aa_types.extend(xh_pocket[:, self.x_dims:].argmax(1).detach().cpu().tolist())
            done += n_b
        ca = self.pocket_representation == 'CA'
        out = {'kl_div_atom_types': self._type_kl(self.dataset_info.get('phar_hist'), self.dataset_info['phar_encoder'], phar_types),
               'kl_div_residue_types': self._type_kl(self.dataset_info.get('aa_hist') if ca else None,
                                                     self.dataset_info['aa_encoder'] if ca else {}, aa_types)}
        print('kl_div_atom_types:', out['kl_div_atom_types'])
        print('kl_div_residue_types:', out['kl_div_residue_types'])
        return out

    # ------------------------------------------------------------------ sampling entry points
    @torch.no_grad()
    def sample_given_batch(self, batch, timesteps=None, **kw):
        """Sampling half of sample_and_analyze_given_pocket (lightning_modules.py:352-373): one collated
        batch dict -> (x list, type list per sample).  The RDKit-based analysis is out of scope."""
        phar, pocket = self.get_phar_and_pocket(batch)
        num_nodes_phar = self.ddpm.size_distribution.sample_conditional(n1=None, n2=pocket['size'])
        xh_phar, xh_pocket, phar_mask, _ = self.ddpm.sample_given_pocket(pocket, num_nodes_phar,
                                                                        timesteps=timesteps, **kw)
        x = xh_phar[:, :self.x_dims].detach().cpu()
        phar_type = xh_phar[:, self.x_dims:].argmax(1).detach().cpu()
        pm = phar_mask.cpu()
        return list(zip(utils.batch_to_list(x, pm), utils.batch_to_list(phar_type, pm)))

    def generate_phars(self, pdb_file, n_samples, pocket_ids=None, ref_ligand=None, num_nodes_phar=None,
                       sanitize=False, largest_frag=False, relax_iter=0, timesteps=None, **kwargs):
        """Generate pharmacophore point clouds inside a pocket (lightning_modules.py:385-541).

        pocket_ids: residues as '<chain>:<resi>'; ref_ligand: '<chain>:<resi>' alternative.
        sanitize / largest_frag / relax_iter and the inpainting kwargs are accepted and have no
        effect in conditional mode, as in the reference (quirk Q10); in mode 'joint' the kwargs
        (resamplings, jump_length) go to EnVariationalDiffusion.inpaint (lightning_modules.py:466-486)."""
        assert (pocket_ids is None) ^ (ref_ligand is None)
        sampler_kw = {k: kwargs.pop(k) for k in ('noise', 'seed') if k in kwargs}
        pdb_struct = utils.parse_pdb(pdb_file)
        if pocket_ids is not None:
            residues = [pdb_struct[x.split(':')[0]][(' ', int(x.split(':')[1]), ' ')] for x in pocket_ids]
        else:
            residues = utils.get_pocket_from_ligand(pdb_struct, ref_ligand)
        if self.pocket_representation == 'CA':
            pocket_coord = torch.tensor(np.array([res['CA'].get_coord() for res in residues]),
                                        device=self.device, dtype=FLOAT_TYPE)
            pocket_types = torch.tensor([self.pocket_type_encoder[utils.three_to_one(res.get_resname())]
                                         for res in residues], device=self.device)
        else:
            atoms = [a for res in residues for a in res.get_atoms()
                     if (a.element.capitalize() in self.pocket_type_encoder or a.element != 'H')]
            pocket_coord = torch.tensor(np.array([a.get_coord() for a in atoms]), device=self.device, dtype=FLOAT_TYPE)
            pocket_types = torch.tensor([self.pocket_type_encoder[a.element.capitalize()] for a in atoms],
                                        device=self.device)     # KeyError for unknown non-H elements (Q12)
        pocket_one_hot = F.one_hot(pocket_types, num_classes=len(self.pocket_type_encoder))
        pocket_size = torch.tensor([len(pocket_coord)] * n_samples, device=self.device, dtype=INT_TYPE)
        pocket_mask = torch.repeat_interleave(torch.arange(n_samples, device=self.device, dtype=INT_TYPE),
                                              len(pocket_coord))
        pocket = {'x': pocket_coord.repeat(n_samples, 1), 'one_hot': pocket_one_hot.repeat(n_samples, 1),
                  'size': pocket_size, 'mask': pocket_mask}
        pocket_com_before = _scatter_mean(pocket['x'], pocket['mask'], n_samples)
        if num_nodes_phar is None:
            num_nodes_phar = self.ddpm.size_distribution.sample_conditional(n1=None, n2=pocket['size'])
        if type(self.ddpm) == EnVariationalDiffusion:
            # inpainting: every pocket node is fixed, every phar node is generated (lightning_modules.py:466-486)
            num_nodes_phar = torch.as_tensor(num_nodes_phar, device=self.device)
            phar_mask = utils.num_nodes_to_batch_mask(len(num_nodes_phar), num_nodes_phar, self.device)
            phar = {'x': torch.zeros((len(phar_mask), self.x_dims), device=self.device, dtype=FLOAT_TYPE),
                    'one_hot': torch.zeros((len(phar_mask), self.phar_nf), device=self.device, dtype=FLOAT_TYPE),
                    'size': num_nodes_phar, 'mask': phar_mask}
            phar_mask_fixed = torch.zeros(len(phar_mask), device=self.device)
            pocket_mask_fixed = torch.ones(len(pocket['mask']), device=self.device)
            xh_phar, xh_pocket, phar_mask, pocket_mask = self.ddpm.inpaint(
                phar, pocket, phar_mask_fixed, pocket_mask_fixed, timesteps=timesteps, **sampler_kw, **kwargs)
        elif isinstance(self.ddpm, ConditionalDDPM):
            xh_phar, xh_pocket, phar_mask, pocket_mask = self.ddpm.sample_given_pocket(
                pocket, num_nodes_phar, timesteps=timesteps, **sampler_kw)
        else:
            raise NotImplementedError
        # move the generated points back to the original pocket position
        pocket_com_after = _scatter_mean(xh_pocket[:, :self.x_dims], pocket_mask, n_samples)
        xh_pocket[:, :self.x_dims] += (pocket_com_before - pocket_com_after)[pocket_mask]
        xh_phar[:, :self.x_dims] += (pocket_com_before - pocket_com_after)[phar_mask]
        phar_mask = phar_mask.cpu()
        x = xh_phar[:, :self.x_dims].detach().cpu()
        phar_type = xh_phar[:, self.x_dims:].argmax(1).detach().cpu()
        # Quirk Q9 kept: the counter restarts for every sample and advances per POINT, so
        # 'Molecule_k' collects the k-th point of all samples, grouped by predicted type.
        phar_to_coords = {}
        for coords_batch, types in zip(utils.batch_to_list(x, phar_mask), utils.batch_to_list(phar_type, phar_mask)):
            names = [self.dataset_info['phar_decoder'][int(t)] for t in types]
            for k, (name, coords) in enumerate(zip(names, coords_batch), start=1):
                phar_to_coords.setdefault(f'Molecule_{k}', {}).setdefault(name, []).append(coords)
        return phar_to_coords
