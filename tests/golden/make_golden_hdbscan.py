#!/usr/bin/env python3
"""Golden vectors for the HDBSCAN path.

The reference calls the third-party `hdbscan` package, which is not installed in this image
and is not pinned by the reference: parity for this row is UNPINNED.  What is recorded here
is the output of the HDBSCAN that IS available offline -- sklearn.cluster.HDBSCAN 1.7.2, a
port of the same algorithm -- on seeded synthetic latents, together with the exact
(float64, brute force) core distances and the weight of the mutual-reachability spanning
tree from scipy.  Run in this container:  python tests/golden/make_golden_hdbscan.py"""
import os

import numpy as np
from scipy.sparse.csgraph import minimum_spanning_tree
from sklearn.cluster import HDBSCAN


def latents(seed, n, d, k):
    rng = np.random.default_rng(seed)
    cents = rng.normal(size=(k, d)) * 3
    parts = [c + rng.normal(size=(n // k, d)) * rng.uniform(0.3, 1.0) for c in cents]
    parts.append(rng.uniform(-8, 8, size=(n // 20, d)))
    X = np.concatenate(parts).astype(np.float32)
    return X[rng.permutation(len(X))]


def main():
    out = {}
    for tag, (seed, n, d, k, mcs, ms) in {"a": (11, 3000, 8, 5, 100, 100), "b": (12, 2000, 4, 4, 50, 20),
                                           "c": (13, 2500, 3, 6, 250, 250)}.items():
        X = latents(seed, n, d, k)
        X64 = X.astype(np.float64)
        D = np.sqrt(((X64[:, None, :] - X64[None, :, :]) ** 2).sum(-1))
        core = np.sort(D, axis=1)[:, ms - 1]
        mr = np.maximum(D, np.maximum(core[:, None], core[None, :]))
        np.fill_diagonal(mr, 0)
        w = minimum_spanning_tree(mr).sum()
        labels = HDBSCAN(min_cluster_size=mcs, min_samples=ms, algorithm="brute").fit_predict(X)
        out[f"{tag}_params"] = np.array([seed, n, d, k, mcs, ms])
        out[f"{tag}_X"] = X
        out[f"{tag}_core"] = core
        out[f"{tag}_mst_weight"] = np.array([w])
        out[f"{tag}_labels"] = labels.astype(np.int32)
        print(tag, X.shape, "clusters", labels.max() + 1, "noise", int((labels < 0).sum()), "mst", w)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "hdbscan.npz"), **out)


if __name__ == "__main__":
    main()
