#!/usr/bin/env python3
"""How often does the pipeline merge two of the eight genomes of the accuracy stand-in?  Runs lrbinner.py
reads on tests/helpers.synth_sim8 with several GC spacings, a few seeds each (3 s per run on the GPU), and
prints bins / F1 -- used to pick a spacing at which the reference's behaviour (and this build's) is the
same in every run, so that the accuracy gate does not depend on a coin.  python scripts/sim8_explore.py"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers

CANDIDATES = {
    "2.0-2.9% steps (round-2 first choice)": (0.40, 0.42, 0.44, 0.46, 0.50, 0.52, 0.56, 0.60),
    "3% steps": (0.38, 0.41, 0.44, 0.47, 0.50, 0.53, 0.57, 0.61),
    "3.5% steps": (0.36, 0.395, 0.43, 0.465, 0.50, 0.535, 0.57, 0.61),
    "4% steps": (0.34, 0.38, 0.42, 0.46, 0.50, 0.54, 0.58, 0.62),
}
for name, gc in CANDIDATES.items():
    helpers.SIM8_GC = gc
    reads, labels = helpers.synth_sim8()
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
        fa = os.path.join(tmp, "r.fasta")
        helpers.write_fasta(fa, reads)
        res = []
        for seed in range(1, 9):
            out = os.path.join(tmp, f"o{seed}")
            cmd = [sys.executable, os.path.join(ROOT, "lrbinner.py"), "reads", "-r", fa, "-o", out, "-k", "3", "-bc", "10",
                   "-bs", "2", "--ae-dims", "4", "--ae-epochs", "200", "-bit", "0", "-mbs", "500", "--cuda", "-t", "8"]
            subprocess.run(cmd, check=True, cwd=ROOT, env=dict(os.environ, LRB_SEED=str(seed)),
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            bins = [int(x) for x in open(os.path.join(out, "bins.txt")).read().split()]
            p, r, f1, nb = helpers.binning_scores(bins, labels)
            res.append((nb, round(f1, 2)))
            subprocess.run(["rm", "-rf", out])
        print(name, gc, "->", res, flush=True)
