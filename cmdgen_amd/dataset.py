"""Processed-complex dataset: NPZ -> per-complex tensors -> flat batch dict
(counterpart of DiffPhar/dataset.py:7-64; NPZ schema written by
process_crossdock_ca_only.py:195-207: names, phar_coords, phar_one_hot, phar_mask,
pocket_c_alpha, pocket_one_hot, pocket_mask).

Host-side numpy/torch only; the batch dict it produces is what
``PharPocketDDPM.get_phar_and_pocket`` consumes (lightning_modules.py:172-186).
"""
from __future__ import annotations

import numpy as np
import torch


class ProcessedLigandPharPocketDataset(torch.utils.data.Dataset):
    def __init__(self, npz_path, center=True):
        with np.load(npz_path, allow_pickle=True) as f:
            raw = {k: f[k] for k in f.files}
        cuts = {'phar': np.where(np.diff(raw['phar_mask']))[0] + 1,
                'pocket': np.where(np.diff(raw['pocket_mask']))[0] + 1}
        self.data = {}
        for k, v in raw.items():
            if k == 'names':
                self.data[k] = v
                continue
            parts = np.split(v, cuts['phar' if 'phar' in k else 'pocket'])
            self.data[k] = [torch.from_numpy(np.ascontiguousarray(p)) for p in parts]
        self.data['num_phar_atoms'] = torch.tensor([len(x) for x in self.data['phar_mask']])
        self.data['num_pocket_nodes'] = torch.tensor([len(x) for x in self.data['pocket_mask']])
        if center:     # joint centre of gravity of phar + pocket nodes of each complex (dataset.py:33-39)
            for i in range(len(self.data['phar_coords'])):
                pc, qc = self.data['phar_coords'][i], self.data['pocket_c_alpha'][i]
                mean = (pc.sum(0) + qc.sum(0)) / (len(pc) + len(qc))
                self.data['phar_coords'][i] = pc - mean
                self.data['pocket_c_alpha'][i] = qc - mean

    def __len__(self):
        return len(self.data['names'])

    def __getitem__(self, idx):
        return {k: v[idx] for k, v in self.data.items()}

    @staticmethod
    def collate_fn(batch):
        """List of complexes -> one flat batch dict (the format PharPocketDDPM.get_phar_and_pocket reads): names as a
        list, node counts as an int tensor, every per-node array concatenated, and the two masks rebuilt as the
        complex's position in THIS batch - float valued, as the reference leaves them (quirk Q13: they are cast to
        int64 on the device later, lightning_modules.py:177, :184)."""
        keys = list(batch[0])
        counts = {'phar_mask': torch.tensor([len(c['phar_mask']) for c in batch]),
                  'pocket_mask': torch.tensor([len(c['pocket_mask']) for c in batch])}
        position = torch.arange(len(batch), dtype=torch.float32)

        def merge(key):
            column = [c[key] for c in batch]
            if key == 'names':
                return column
            if key in ('num_phar_atoms', 'num_pocket_nodes'):
                return torch.tensor(column)
            if key in counts:
                return torch.repeat_interleave(position, counts[key])
            if 'mask' in key:
                return torch.repeat_interleave(position, torch.tensor([len(v) for v in column]))
            return torch.cat(column, dim=0)
        return {key: merge(key) for key in keys}


def write_synthetic_npz(path, n_complexes=6, seed=0, representation='CA'):
    """A processed-dataset file in the reference's schema, filled with synthetic complexes (tests, demos)."""
    from .synthetic import make_pockets
    rng = np.random.Generator(np.random.PCG64(seed))
    pb = make_pockets(n_complexes, representation, ragged=True, first_index=10 * seed)
    shift = rng.normal(size=(n_complexes, 3)).astype(np.float32) * 20.0         # un-centred, like raw PDB frames
    phar_xyz, phar_oh, phar_mask = [], [], []
    for b in range(n_complexes):
        nl = int(pb.num_nodes_phar[b])
        com = pb.x[pb.mask == b].mean(0)
        phar_xyz.append((com + rng.normal(size=(nl, 3)) * 2.0 + shift[b]).astype(np.float32))
        phar_oh.append(np.eye(8, dtype=np.float32)[rng.integers(0, 8, size=nl)])
        phar_mask.append(np.full(nl, b))
    np.savez(path, names=np.array([f'complex_{b}' for b in range(n_complexes)]),
             phar_coords=np.concatenate(phar_xyz), phar_one_hot=np.concatenate(phar_oh),
             phar_mask=np.concatenate(phar_mask), pocket_c_alpha=pb.x + shift[pb.mask],
             pocket_one_hot=pb.one_hot, pocket_mask=pb.mask)
    return pb
