#!/usr/bin/env python3
"""sklearn's labels for 200 k fragment latents of a C5 run.

The reference's `hdbscan` package is not installable in this image: the row stays PARITY UNPINNED.  What can be
held is label identity with the implementation of the same published algorithm that IS here,
sklearn.cluster.HDBSCAN 1.7.2, at a size where border effects would show: the first 200,000 rows of latent.npy of a
`lrbinner.py contigs` run on the MI355X (scripts/c5_full.py with C5_SAVE_LATENT; 1.66 M fragments, 8 dims, 101
clusters) -- min_cluster_size 250 as cluster_utils.py:489-494 calls it, everything else at its default.
Build container only (minutes of CPU):

    python tests/golden/make_golden_hdbscan_c5.py gpurun_out/c5_latent_200k.npy
    python tests/golden/make_golden_hdbscan_c5.py            # again from the X the fixture already holds

Writes tests/golden/hdbscan_c5_200k.npz: X (float32, as the kernels see it), labels (sklearn's, its defaults:
min_samples = min_cluster_size = 250 with the point itself counted), labels_ms251 (sklearn with min_samples = 251:
the 250-th OTHER point -- the convention of the hdbscan package's Boruvka paths, which is the library's DEFAULT
(core_excludes_self=None), so that the shipped default is the compared one), params."""
import os
import sys
import time

import numpy as np
from sklearn.cluster import HDBSCAN
import sklearn


def fit(X, **kw):
    t0 = time.time()
    labels = HDBSCAN(min_cluster_size=250, algorithm="kd_tree", copy=True, **kw).fit_predict(X.astype(np.float64))
    print(f"sklearn {sklearn.__version__} {kw}: {len(X)} points, {labels.max() + 1} clusters, {(labels < 0).sum()} noise, "
          f"{time.time() - t0:.0f} s", flush=True)
    return labels.astype(np.int16)


def main(path=None):
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hdbscan_c5_200k.npz")
    have = dict(np.load(out)) if os.path.exists(out) else {}
    X = np.load(path).astype(np.float32) if path else have["X"]
    if path or "labels" not in have:
        have["labels"] = fit(X)
    have["labels_ms251"] = fit(X, min_samples=251)
    have.update(X=X, min_cluster_size=np.array([250]), sklearn_version=np.array([sklearn.__version__]))
    np.savez_compressed(out, **have)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
