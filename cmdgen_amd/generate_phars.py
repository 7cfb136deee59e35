"""Driver: checkpoint + PDB pocket -> sampled pharmacophore points as JSON
(counterpart of DiffPhar/generate_phars.py:11-66; same flags).

    python -m cmdgen_amd.generate_phars ckpt --pdbfile p.pdb --resi_list A:1 A:2 --n_samples 20

The reference writes to a hard-coded file name in the working directory (quirk Q10); here the
same default name is used unless --outdir is given, in which case the file goes there.
"""
import argparse
import json
from pathlib import Path

import torch

from .lightning_modules import PharPocketDDPM

DEFAULT_JSON = 'phar_to_coords_no_tensor_PI3K_dul.json'


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('checkpoint', type=Path)
    p.add_argument('--pdbfile', type=str)
    p.add_argument('--resi_list', type=str, nargs='+', default=None)
    p.add_argument('--ref_ligand', type=str, default=None)
    p.add_argument('--outdir', type=Path)
    p.add_argument('--n_samples', type=int, default=20)
    p.add_argument('--num_nodes_phar', type=int, default=3)
    p.add_argument('--all_frags', action='store_true')
    p.add_argument('--sanitize', action='store_true')
    p.add_argument('--relax', action='store_true')
    p.add_argument('--resamplings', type=int, default=10)
    p.add_argument('--jump_length', type=int, default=1)
    p.add_argument('--timesteps', type=int, default=None)
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    device = 'cuda' if torch.cuda.is_available() else 'cpu'
    model = PharPocketDDPM.load_from_checkpoint(args.checkpoint, map_location=device).to(device)
    num_nodes_phar = torch.ones(args.n_samples, dtype=int) * args.num_nodes_phar \
        if args.num_nodes_phar is not None else None
    phar_to_coords = model.generate_phars(
        args.pdbfile, args.n_samples, args.resi_list, args.ref_ligand, num_nodes_phar, args.sanitize,
        largest_frag=not args.all_frags, relax_iter=(200 if args.relax else 0),
        resamplings=args.resamplings, jump_length=args.jump_length, timesteps=args.timesteps)
    plain = {mol: {ftype: [c.tolist() for c in coords] for ftype, coords in feats.items()}
             for mol, feats in phar_to_coords.items()}
    out = Path(args.outdir, DEFAULT_JSON) if args.outdir else Path(DEFAULT_JSON)
    out.parent.mkdir(parents=True, exist_ok=True)
    with open(out, 'w') as f:
        json.dump(plain, f)
    print(phar_to_coords)
    return plain


if __name__ == '__main__':
    main()
