#!/usr/bin/env python3
"""How often does a whole `lrbinner.py reads` run on the 432 k-read stand-in end with all eight genomes apart?
One data set, N runs with LRB_SEED = 1..N: bins, precision / recall / F1, wall time per run.
python scripts/e2e_seeds.py [n_runs] [n_reads]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import binning_scores, synth_sim8, write_fasta

n_runs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 432_333
reads, origin = synth_sim8(scale=n_reads / 40350.0)
with tempfile.TemporaryDirectory(dir="/dev/shm") as tmp:
    fa = os.path.join(tmp, "reads.fasta")
    write_fasta(fa, reads)
    del reads
    for seed in range(1, n_runs + 1):
        out = os.path.join(tmp, f"out{seed}")
        cmd = [sys.executable, os.path.join(ROOT, "lrbinner.py"), "reads", "-r", fa, "-o", out, "-k", "3", "-bc", "10",
               "-bs", "2", "--ae-dims", "4", "--ae-epochs", "200", "-bit", "0", "-mbs", "5000", "--cuda", "-t", "16"]
        t0 = time.time()
        subprocess.run(cmd, check=True, cwd=ROOT, env=dict(os.environ, LRB_SEED=str(seed)),
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dt = time.time() - t0
        bins = [int(x) for x in open(os.path.join(out, "bins.txt")).read().split()]
        p, r, f1, nb = binning_scores(bins, origin)
        print(f"seed {seed}: {nb} bins  P {p:.2f}  R {r:.2f}  F1 {f1:.2f}  {dt:.1f} s", flush=True)
        subprocess.run(["rm", "-rf", out])
