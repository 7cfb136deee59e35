"""Consensus pharmacophore from generated point clouds: the step between ``generate_phars`` and GCPG.

Counterpart of ``get_phar/GMM_json.py`` (a top-level script with hard-coded paths in the reference): the
``phar_to_coords`` JSON written by ``generate_phars`` is pooled, clustered with a Gaussian mixture
(scikit-learn, as in the reference: ``GaussianMixture(n_components=7, random_state=42)``), every cluster is typed
by its most probable feature, and the typed centres are written as ``.posp`` lines ``<TYPE> x y z`` - the wire
format ``GCPG/utils/file_utils.py:67-102`` (``load_pp_file``) reads.  Host-side post-processing of ~10^2 points;
nothing here touches the GPU.

    python -m cmdgen_amd.get_phar phar_to_coords.json --n_clusters 7 --out output.posp
"""
from __future__ import annotations

import argparse
import json
from pathlib import Path
from typing import Dict, List

import numpy as np

# GMM_json.py:116-134
MAPPING = {'Aromatic': 1, 'Hydrophobe': 2, 'PosIonizable': 3, 'Acceptor': 4, 'Donor': 5, 'LumpedHydrophobe': 6, 'others': 7}
IDX2PHAR = {1: 'AROM', 2: 'HYBL', 3: 'POSC', 4: 'HACC', 5: 'HDON', 6: 'LHYBL', 7: 'UNKNOWN'}


def gmm_consensus(phar_to_coords: Dict[str, Dict[str, list]], n_clusters: int = 7, random_state: int = 42):
    """-> {'Cluster k': {'Center Coordinates': (x, y, z), 'Most Probable Feature': name}} (GMM_json.py:16-113).

    Per feature the responsibilities of its points are summed per cluster and normalised over clusters
    (:41-55); a cluster's type is the feature with the largest normalised share there (first in insertion order
    on ties, as ``sorted(..., reverse=True)`` is stable)."""
    from sklearn.mixture import GaussianMixture
    vectors = []
    for features in phar_to_coords.values():
        for coordinates in features.values():
            vectors.extend(coordinates)
    X = np.array(vectors)
    gmm = GaussianMixture(n_components=n_clusters, random_state=random_state)
    gmm.fit(X)
    centers = gmm.means_
    feature_probs = {f: np.zeros(n_clusters) for feats in phar_to_coords.values() for f in feats}
    for features in phar_to_coords.values():
        for feature, coordinates in features.items():
            feature_probs[feature] += np.sum(gmm.predict_proba(coordinates), axis=0)
    for feature in feature_probs:
        feature_probs[feature] /= np.sum(feature_probs[feature])
    out = {}
    for i in range(n_clusters):
        ranked = sorted(feature_probs.keys(), key=lambda f: feature_probs[f][i], reverse=True)
        out[f'Cluster {i + 1}'] = {'Center Coordinates': (centers[i, 0], centers[i, 1], centers[i, 2]),
                                   'Most Probable Feature': ranked[0]}
    return out


def posp_lines(cluster_data) -> List[str]:
    """GMM_json.py:136-147; clusters typed with a feature outside the mapping ('NegIonizable') are dropped, as there."""
    lines = []
    for data in cluster_data.values():
        feature = data['Most Probable Feature']
        if feature in MAPPING:
            c = data['Center Coordinates']
            lines.append(f"{IDX2PHAR[MAPPING[feature]]} {c[0]:.2f} {c[1]:.2f} {c[2]:.2f}")
    return lines


def write_posp(path, cluster_data) -> List[str]:
    lines = posp_lines(cluster_data)
    with open(path, 'w') as f:          # :150-153
        for line in lines:
            f.write(line + '\n')
    return lines


def main(argv=None):
    p = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    p.add_argument('json', type=Path, help='phar_to_coords JSON written by generate_phars')
    p.add_argument('--n_clusters', type=int, default=7)
    p.add_argument('--random_state', type=int, default=42)
    p.add_argument('--out', type=Path, default=Path('output.posp'))
    a = p.parse_args(argv)
    with open(a.json) as f:
        data = json.load(f)
    clusters = gmm_consensus(data, a.n_clusters, a.random_state)
    lines = write_posp(a.out, clusters)
    print('\n'.join(lines))
    return lines


if __name__ == '__main__':
    main()
