#!/usr/bin/env python3
"""Geometry of the latents of the 40 k-read accuracy stand-in: the reference's (tests/golden/sim8_ref_s*.npz) next to
this build's (trained here on the GPU, LRB_SEED 1..3): per genome the mean distance of a read to its genome's
centroid, and the distances between the centroids of neighbouring genomes, in the normalised space the cluster search
works in.  python scripts/latent_geometry.py"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import golden_path, synth_sim8, write_fasta


def geometry(lat, origin):
    m = lat / (np.linalg.norm(lat, axis=1, keepdims=True) * np.sqrt(2))
    cen = np.stack([m[origin == g].mean(0) for g in range(8)])
    spread = np.array([np.sqrt(((m[origin == g] - cen[g]) ** 2).sum(1)).mean() for g in range(8)])
    d = np.sqrt(((cen[:, None] - cen[None]) ** 2).sum(-1))
    nb = np.array([d[g, g + 1] for g in range(7)])
    return spread, nb


reads, origin = synth_sim8()
origin = np.asarray(origin)
rows = []
for s in (1, 2, 3):
    z = np.load(golden_path(f"sim8_ref_s{s}.npz"))
    rows.append((f"reference seed {s}", *geometry(z["latent"], origin)))
with tempfile.TemporaryDirectory(dir="/dev/shm") as tmp:
    fa = os.path.join(tmp, "r.fasta")
    write_fasta(fa, reads)
    for s in (1, 2, 3):
        out = os.path.join(tmp, f"o{s}")
        cmd = [sys.executable, os.path.join(ROOT, "lrbinner.py"), "reads", "-r", fa, "-o", out, "-k", "3", "-bc", "10",
               "-bs", "2", "--ae-dims", "4", "--ae-epochs", "200", "-bit", "0", "-mbs", "500", "--cuda", "-t", "8"]
        subprocess.run(cmd, check=True, cwd=ROOT, env=dict(os.environ, LRB_SEED=str(s)), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        rows.append((f"this build seed {s}", *geometry(np.load(os.path.join(out, "latent.npy")), origin)))
for name, spread, nb in rows:
    print(f"{name:18s} spread {np.round(spread, 3)} mean {spread.mean():.4f} | neighbour centroids {np.round(nb, 3)} min {nb.min():.3f} | min gap / spread {(nb / (spread[:-1] + spread[1:])).min():.2f}")
