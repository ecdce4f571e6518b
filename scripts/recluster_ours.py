#!/usr/bin/env python3
"""This build's cluster search on a given latent.npy under random.seed(1..N): cluster sizes per seed (the GPU half of
tests/golden/ref_recluster.py).  python scripts/recluster_ours.py LATENT.npy [n_seeds] [min_cluster_size]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lrbinner_amd import cluster_utils

lat = np.load(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
mbs = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
for s in range(1, n + 1):
    random.seed(s)
    cl = cluster_utils.cluster_points(lat, 0, mbs)
    print(f"seed {s}: " + " ".join(str(len(v)) for v in cl.values()), flush=True)
