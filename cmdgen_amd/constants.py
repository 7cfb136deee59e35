"""Dtypes and vocabularies of the DiffPhar path (counterpart of DiffPhar/constants.py:8-9, :96-123).

Only what the denoising path touches is kept: dtypes and the encoder/decoder tables plus
type histograms of the two datasets.  Bond tables, colours and radii are out of scope.
"""
import torch

FLOAT_TYPE = torch.float32      # constants.py:8
INT_TYPE = torch.int64          # constants.py:9

_PHAR = ['Aromatic', 'Hydrophobe', 'PosIonizable', 'NegIonizable', 'Acceptor', 'Donor',
         'LumpedHydrophobe', 'others']
_PHAR_COUNTS = [176393, 329938, 38876, 28234, 485363, 303290, 124515, 30892]
_ATOMS = ['C', 'N', 'O', 'S', 'B', 'Br', 'Cl', 'P', 'I', 'F']
_AA = list('ACDEFGHIKLMNPQRSTVWY')
_AA_COUNTS = [277175, 92406, 254046, 201833, 234995, 376966, 147704, 290683, 173210, 421883,
              157813, 174241, 148581, 120232, 173848, 274430, 247605, 326134, 88552, 226668]
_FULL_ATOM_COUNTS = [23481798, 6139100, 6753114, 278864, 0, 0, 0, 0, 0, 0, 0]


def _enc(names):
    return {n: i for i, n in enumerate(names)}


dataset_params = {
    # full-atom pockets: the residue vocabulary is the 11-way element table (constants.py:96-108)
    'crossdock_full': {
        'atom_encoder': _enc(_ATOMS + ['others']), 'atom_decoder': _ATOMS + ['others'],
        'phar_encoder': _enc(_PHAR), 'phar_decoder': list(_PHAR),
        'aa_encoder': _enc(_ATOMS + ['others']), 'aa_decoder': _ATOMS + ['others'],
        'phar_hist': dict(zip(_PHAR, _PHAR_COUNTS)),
        'aa_hist': dict(zip(_ATOMS + ['others'], _FULL_ATOM_COUNTS)),
    },
    # C-alpha pockets: 20 amino acids (constants.py:110-123)
    'crossdock': {
        'atom_encoder': _enc(_ATOMS), 'atom_decoder': list(_ATOMS),
        'phar_encoder': _enc(_PHAR), 'phar_decoder': list(_PHAR),
        'aa_encoder': _enc(_AA), 'aa_decoder': list(_AA),
        'phar_hist': dict(zip(_PHAR, _PHAR_COUNTS)),
        'aa_hist': dict(zip(_AA, _AA_COUNTS)),
    },
}
