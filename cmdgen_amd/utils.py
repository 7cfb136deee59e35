"""Batch-mask helpers, running queue and a small PDB reader for the driver.

Counterparts: utils.num_nodes_to_batch_mask (utils.py:137-145), batch_to_list (:122-134),
Queue (:12-29), get_pocket_from_ligand (:102-119).  BioPython is not a dependency: the
ATOM/HETATM reader below provides the few accessors generate_phars uses
(lightning_modules.py:410-437).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import List

import numpy as np
import torch


class Queue:
    """Fixed-length history of gradient norms (utils.py:12-29)."""
    def __init__(self, max_len=50):
        self.items: List[float] = []
        self.max_len = max_len

    def __len__(self):
        return len(self.items)

    def add(self, item):
        self.items.insert(0, item)
        if len(self) > self.max_len:
            self.items.pop()

    def mean(self):
        return np.mean(self.items)

    def std(self):
        return np.std(self.items)


def num_nodes_to_batch_mask(n_samples, num_nodes, device):
    assert isinstance(num_nodes, int) or len(num_nodes) == n_samples
    if isinstance(num_nodes, torch.Tensor):
        num_nodes = num_nodes.to(device)
    return torch.repeat_interleave(torch.arange(n_samples, device=device), num_nodes)


def batch_to_list(data, batch_mask):
    order = torch.argsort(batch_mask)          # stable enough: masks are ascending already
    batch_mask = batch_mask[order]
    data = data[order]
    sizes = torch.unique(batch_mask, return_counts=True)[1].tolist()
    return torch.split(data, sizes)


def sizes_from_mask(mask: torch.Tensor, batch: int) -> np.ndarray:
    """Per-sample node counts of an ascending, contiguous batch mask (host array)."""
    m = mask.detach().to('cpu', torch.int64)
    if m.numel() and bool((m[1:] < m[:-1]).any()):
        raise ValueError('batch masks must be ascending and contiguous (as every mask the reference builds is)')
    return torch.bincount(m, minlength=batch).numpy().astype(np.int64)


# ---------------------------------------------------------------------------- PDB reader
_THREE_TO_ONE = {'ALA': 'A', 'CYS': 'C', 'ASP': 'D', 'GLU': 'E', 'PHE': 'F', 'GLY': 'G', 'HIS': 'H',
                 'ILE': 'I', 'LYS': 'K', 'LEU': 'L', 'MET': 'M', 'ASN': 'N', 'PRO': 'P', 'GLN': 'Q',
                 'ARG': 'R', 'SER': 'S', 'THR': 'T', 'VAL': 'V', 'TRP': 'W', 'TYR': 'Y'}


def three_to_one(resname: str) -> str:
    return _THREE_TO_ONE[resname.strip().upper()]     # KeyError for non-standard names, like Bio's


def is_aa(resname: str, standard: bool = True) -> bool:
    return resname.strip().upper() in _THREE_TO_ONE


class Atom:
    def __init__(self, name, element, coord):
        self.name, self.element, self.coord = name, element, coord

    def get_coord(self):
        return self.coord


class Residue:
    def __init__(self, resname, rid):
        self.resname, self.id = resname, rid       # id = (hetflag, resseq, icode) as Bio.PDB
        self.atoms: "OrderedDict[str, Atom]" = OrderedDict()

    def get_resname(self):
        return self.resname

    def get_atoms(self):
        return list(self.atoms.values())

    def __getitem__(self, name):
        return self.atoms[name]


class Chain:
    def __init__(self, cid):
        self.id = cid
        self.residues: "OrderedDict[tuple, Residue]" = OrderedDict()

    def __getitem__(self, rid):
        return self.residues[rid]

    def get_residues(self):
        return list(self.residues.values())


class Model:
    def __init__(self):
        self.chains: "OrderedDict[str, Chain]" = OrderedDict()

    def __getitem__(self, cid):
        return self.chains[cid]

    def get_residues(self):
        return [r for c in self.chains.values() for r in c.get_residues()]


def parse_pdb(path: str) -> Model:
    """First model of a PDB file (what PDBParser(QUIET=True).get_structure('', f)[0] yields)."""
    model = Model()
    with open(path) as f:
        for line in f:
            rec = line[:6]
            if rec.startswith('ENDMDL'):
                break
            if rec not in ('ATOM  ', 'HETATM'):
                continue
            name = line[12:16].strip()
            altloc = line[16]
            if altloc not in (' ', 'A'):
                continue
            resname = line[17:20].strip()
            cid = line[21]
            resseq = int(line[22:26])
            icode = line[26]
            xyz = np.array([float(line[30:38]), float(line[38:46]), float(line[46:54])], dtype=np.float32)
            element = line[76:78].strip() if len(line) >= 78 else ''
            if not element:
                element = ''.join(ch for ch in name if ch.isalpha())[:1]
            het = ' ' if rec == 'ATOM  ' else ('W' if resname == 'HOH' else 'H_' + resname)
            chain = model.chains.setdefault(cid, Chain(cid))
            res = chain.residues.setdefault((het, resseq, icode), Residue(resname, (het, resseq, icode)))
            if name not in res.atoms:
                res.atoms[name] = Atom(name, element.upper(), xyz)
    return model


def get_pocket_from_ligand(pdb_model: Model, ligand_id: str, dist_cutoff: float = 8.0):
    """Residues with any atom within dist_cutoff of the ligand `chain:resi` (utils.py:102-119).

    Quirk kept: the reference compares residue.id[1] with the *string* resi, which never
    matches, so the ligand itself is excluded only by the standard-amino-acid filter."""
    chain, resi = ligand_id.split(':')
    lig = [r for r in pdb_model[chain].get_residues() if r.id[1] == int(resi)]
    assert len(lig) == 1
    lig_xyz = torch.from_numpy(np.array([a.get_coord() for a in lig[0].get_atoms()]))
    out = []
    for res in pdb_model.get_residues():
        if res.id[1] == resi:
            continue
        xyz = torch.from_numpy(np.array([a.get_coord() for a in res.get_atoms()]))
        if is_aa(res.get_resname(), standard=True) and torch.cdist(xyz, lig_xyz).min() < dist_cutoff:
            out.append(res)
    return out
