#!/usr/bin/env python3
"""HDBSCAN on the GPU vs sklearn on the host: parity (core distances, spanning-tree weight,
labels) and timing.  python scripts/hdb_probe.py [n ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from lrbinner_amd import device as lrb

def blobs(n, d, seed=0, k=8):
    rng = np.random.default_rng(seed)
    cents = rng.normal(size=(k, d)) * 3
    parts = [c + rng.normal(size=(n // k, d)) * rng.uniform(0.3, 1.0) for c in cents]
    parts.append(rng.uniform(-8, 8, size=(n // 20, d)))
    return np.concatenate(parts).astype(np.float32)

ctx = lrb.Context(0, use_torch_stream=True)
dev = torch.device("cuda", 0)
sizes = [int(a) for a in sys.argv[1:]] or [5000, 50000]
for n in sizes:
    for d, mcs in ((8, 250), (4, 100)):
        X = blobs(n, d)
        N = len(X)
        Xt = torch.from_numpy(X).to(dev)
        torch.cuda.synchronize()
        t0 = time.time(); core = ctx.hdb_core_dist_dev(Xt, mcs); torch.cuda.synchronize(); t_core = time.time() - t0
        t0 = time.time(); u, v, w, rounds = ctx.hdb_mst_dev(Xt, core); t_mst = time.time() - t0
        t0 = time.time(); lab, nc = lrb.hdb_labels(N, u, v, w, mcs); t_lab = time.time() - t0
        msg = f"n={N} d={d} mcs={mcs}: core {t_core*1e3:.1f} ms, mst {t_mst*1e3:.1f} ms ({rounds} rounds), labels {t_lab*1e3:.1f} ms, clusters {nc}"
        if N <= 60000:
            from sklearn.cluster import HDBSCAN
            from sklearn.neighbors import NearestNeighbors
            from sklearn.metrics import adjusted_rand_score
            t0 = time.time(); ref = HDBSCAN(min_cluster_size=mcs).fit_predict(X); t_ref = time.time() - t0
            nd, _ = NearestNeighbors(n_neighbors=mcs).fit(X).kneighbors(X, mcs)
            err = np.abs(core.cpu().numpy() - nd[:, -1]).max()
            msg += f" | sklearn {t_ref:.2f} s, core max abs err {err:.2e}, ARI {adjusted_rand_score(ref, lab):.4f}, mst weight {w.astype(np.float64).sum():.4f}"
        print(msg, flush=True)
