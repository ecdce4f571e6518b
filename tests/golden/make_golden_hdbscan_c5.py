#!/usr/bin/env python3
"""sklearn's labels for 200 k fragment latents of a C5 run.

The reference's `hdbscan` package is not installable in this image: the row stays PARITY UNPINNED.  What can be
held is label identity with the implementation of the same published algorithm that IS here,
sklearn.cluster.HDBSCAN 1.7.2, at a size where border effects would show: the first 200,000 rows of latent.npy of a
`lrbinner.py contigs` run on the MI355X (scripts/c5_full.py with C5_SAVE_LATENT; 1.66 M fragments, 8 dims, 101
clusters) -- min_cluster_size 250 as cluster_utils.py:489-494 calls it, everything else at its default.
Build container only (minutes of CPU):

    python tests/golden/make_golden_hdbscan_c5.py gpurun_out/c5_latent_200k.npy

Writes tests/golden/hdbscan_c5_200k.npz: X (float32, as the kernels see it), labels (sklearn's), params."""
import os
import sys
import time

import numpy as np
from sklearn.cluster import HDBSCAN
import sklearn


def main(path):
    X = np.load(path).astype(np.float32)
    t0 = time.time()
    labels = HDBSCAN(min_cluster_size=250, algorithm="kd_tree", copy=True).fit_predict(X.astype(np.float64))
    print(f"sklearn {sklearn.__version__}: {len(X)} points, {labels.max() + 1} clusters, {(labels < 0).sum()} noise, "
          f"{time.time() - t0:.0f} s", flush=True)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hdbscan_c5_200k.npz")
    np.savez_compressed(out, X=X, labels=labels.astype(np.int16), min_cluster_size=np.array([250]),
                        sklearn_version=np.array([sklearn.__version__]))


if __name__ == "__main__":
    main(sys.argv[1])
