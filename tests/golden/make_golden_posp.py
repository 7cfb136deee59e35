"""Golden for the consensus-pharmacophore step that follows generate_phars in the reference pipeline:
get_phar/GMM_json.py (JSON of generated points -> 7-component GMM -> one typed centre per cluster -> .posp, the
wire format GCPG/utils/file_utils.py:67-102 reads).  The reference file is a top-level script with hard-coded
file names; it is executed unmodified with runpy in a scratch directory (matplotlib on the Agg backend).
Writes tests/golden/g10_posp.json = {"cases": [{"input": <phar_to_coords>, "posp": <text>}]}.

    python tests/golden/make_golden_posp.py
"""
import contextlib
import io
import json
import os
import runpy
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SCRIPT = '/root/reference/DiffPhar/get_phar/GMM_json.py'
TYPES = ['Aromatic', 'Hydrophobe', 'PosIonizable', 'NegIonizable', 'Acceptor', 'Donor', 'LumpedHydrophobe', 'others']


def synthetic_points(seed, n_samples, n_points, spread):
    """generate_phars-shaped output (quirk Q9 grouping: 'Molecule_k' = k-th point of every sample): n_points
    true sites, each with a dominant type; every sample places its k-th point near site k."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sites = rng.uniform(-8, 8, size=(n_points, 3))
    dom = rng.integers(0, len(TYPES), size=n_points)
    out = {}
    for k in range(n_points):
        feats = {}
        for _ in range(n_samples):
            t = dom[k] if rng.uniform() < 0.8 else rng.integers(0, len(TYPES))
            p = sites[k] + rng.normal(size=3) * spread
            feats.setdefault(TYPES[int(t)], []).append([float(v) for v in p])
        out[f'Molecule_{k + 1}'] = feats
    return out


def run_reference(data):
    os.environ['MPLBACKEND'] = 'Agg'
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as d:
        os.chdir(d)
        try:
            with open('phar_to_coords_no_tensor_PARP1.json', 'w') as f:
                json.dump(data, f)
            with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                runpy.run_path(SCRIPT, run_name='__main__')
            return open('output.posp').read()
        finally:
            os.chdir(cwd)


def main():
    cases = []
    for seed, n_samples, n_points, spread in [(1, 20, 7, 0.5), (2, 30, 9, 0.8), (3, 12, 7, 1.5)]:
        data = synthetic_points(seed, n_samples, n_points, spread)
        posp = run_reference(data)
        cases.append({'input': data, 'posp': posp})
        print(f'seed {seed}: {len(posp.splitlines())} lines\n{posp}')
    with open(os.path.join(HERE, 'g10_posp.json'), 'w') as f:
        json.dump({'cases': cases}, f)


if __name__ == '__main__':
    main()
