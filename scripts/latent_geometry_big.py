#!/usr/bin/env python3
"""The 432 k-read stand-in exactly as the reference harness makes it (synth_sim8(scale=432331/40350)): this build's
runs under LRB_SEED 1..3 -- latent geometry (as scripts/latent_geometry.py), reads left to the likelihood assignment,
and where the wrongly binned reads come from (cluster members / left-overs).  The same numbers for a REFERENCE run:
python scripts/latent_geometry_big.py --dir WORK  (WORK = the harness's /dev/shm/sim8_big: labels.npy + out/)."""
import os, pickle, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def report(name, out, origin):
    lat = np.load(os.path.join(out, "latent.npy"))
    m = lat / (np.linalg.norm(lat, axis=1, keepdims=True) * np.sqrt(2))
    cen = np.stack([m[origin == g].mean(0) for g in range(8)])
    spread = np.array([np.sqrt(((m[origin == g] - cen[g]) ** 2).sum(1)).mean() for g in range(8)])
    d = np.sqrt(((cen[:, None] - cen[None]) ** 2).sum(-1))
    nb = np.array([d[g, g + 1] for g in range(7)])
    bins = np.array([int(x) for x in open(os.path.join(out, "bins.txt")).read().split()])
    # majority genome of every bin; a read is wrong when it is not of its bin's majority genome
    wrong = np.zeros(len(bins), bool)
    for b in np.unique(bins):
        idx = np.flatnonzero(bins == b)
        wrong[idx] = origin[idx] != np.bincount(origin[idx]).argmax()
    log = open(os.path.join(out, "LRBinner.log")).read() if os.path.exists(os.path.join(out, "LRBinner.log")) else ""
    left = re.findall(r"Unclassified points to cluster (\d+)", log)
    print(f"{name}: bins {len(np.unique(bins))}  wrong reads {int(wrong.sum())} ({100 * wrong.mean():.3f} %)  left to the "
          f"likelihood assignment {left[-1] if left else '?'}  spread mean {spread.mean():.4f} {np.round(spread, 3)}  "
          f"neighbour centroids min {nb.min():.3f} {np.round(nb, 3)}", flush=True)


if len(sys.argv) > 2 and sys.argv[1] == "--dir":
    work = sys.argv[2]
    report("reference run in " + work, os.path.join(work, "out"), np.load(os.path.join(work, "labels.npy")))
    sys.exit(0)
from helpers import synth_sim8, write_fasta
reads, origin = synth_sim8(scale=432331 / 40350.0)
origin = np.asarray(origin)
with tempfile.TemporaryDirectory(dir="/dev/shm") as tmp:
    fa = os.path.join(tmp, "reads.fasta")
    write_fasta(fa, reads)
    del reads
    for s in (1, 2, 3, 4):
        out = os.path.join(tmp, f"o{s}")
        cmd = [sys.executable, os.path.join(ROOT, "lrbinner.py"), "reads", "-r", fa, "-o", out, "-k", "3", "-bc", "10",
               "-bs", "2", "--ae-dims", "4", "--ae-epochs", "200", "-bit", "0", "-mbs", "5000", "--cuda", "-t", "16"]
        subprocess.run(cmd, check=True, cwd=ROOT, env=dict(os.environ, LRB_SEED=str(s)), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        report(f"this build, LRB_SEED {s}", out, origin)
