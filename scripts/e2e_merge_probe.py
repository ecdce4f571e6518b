#!/usr/bin/env python3
"""A run of the 432 k-read stand-in that ends with seven bins: is it the latents or the cluster search?  Trains once
with LRB_SEED (default 2: a merging seed in profiles/r02_e2e_seeds.txt), then clusters the same latent.npy under
twenty different random seeds and reports bins / F1; also the distance between the latent centroids of the genomes.
python scripts/e2e_merge_probe.py [seed]"""
import os, random, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import binning_scores, synth_sim8, write_fasta

seed = sys.argv[1] if len(sys.argv) > 1 else "2"
reads, origin = synth_sim8(scale=432_333 / 40350.0)
origin = np.asarray(origin)
with tempfile.TemporaryDirectory(dir="/dev/shm") as tmp:
    fa = os.path.join(tmp, "reads.fasta")
    write_fasta(fa, reads)
    del reads
    out = os.path.join(tmp, "out")
    cmd = [sys.executable, os.path.join(ROOT, "lrbinner.py"), "reads", "-r", fa, "-o", out, "-k", "3", "-bc", "10",
           "-bs", "2", "--ae-dims", "4", "--ae-epochs", "200", "-bit", "0", "-mbs", "5000", "--cuda", "-t", "16"]
    subprocess.run(cmd, check=True, cwd=ROOT, env=dict(os.environ, LRB_SEED=seed), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    bins = [int(x) for x in open(os.path.join(out, "bins.txt")).read().split()]
    print("the run itself:", binning_scores(bins, origin))
    lat = np.load(os.path.join(out, "latent.npy"))
    # genome centroids in the normalised latent space the clustering works in
    m = lat / (np.linalg.norm(lat, axis=1, keepdims=True) * np.sqrt(2))
    cen = np.stack([m[origin == g].mean(0) for g in range(8)])
    spread = np.array([np.sqrt(((m[origin == g] - cen[g]) ** 2).sum(1)).mean() for g in range(8)])
    d = np.sqrt(((cen[:, None] - cen[None]) ** 2).sum(-1))
    print("mean distance of a read to its genome's centroid:", np.round(spread, 4))
    print("centroid distances (neighbours in GC):", np.round([d[g, g + 1] for g in range(7)], 4))
    from lrbinner_amd import cluster_utils
    if os.environ.get("MERGE_PROBE_SAVE"):   # the latents + this build's clusters per seed, for the reference's search on the same latents
        np.save(os.path.join(os.environ["MERGE_PROBE_SAVE"], "merge_probe_latent.npy"), lat)
        with open(os.path.join(os.environ["MERGE_PROBE_SAVE"], "merge_probe_clusters.txt"), "w") as f:
            for s in range(1, 9):
                random.seed(s)
                cl = cluster_utils.cluster_points(lat, 0, 5000)
                f.write(f"seed {s}: " + " ".join(str(len(v)) for v in cl.values()) + "\n")
    for s in range(1, 21):
        random.seed(s)
        o = os.path.join(tmp, f"re{s}")
        os.makedirs(o + "/profiles")
        for f in ("latent.npy", "profiles/com_profs.npy", "profiles/cov_profs.npy"):
            os.symlink(os.path.join(out, f), os.path.join(o, f))
        cluster_utils.perform_binning(o, 0, 5000, False, fa)
        b = [int(x) for x in open(os.path.join(o, "bins.txt")).read().split()]
        p, r, f1, nb = binning_scores(b, origin)
        print(f"clustering seed {s}: {nb} bins F1 {f1:.2f}", flush=True)
