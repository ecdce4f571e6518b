#!/usr/bin/env python3
"""Which C1-sized stand-in strains the method?  helpers.synth_sim8_c1_hard variants through THIS build's pipeline on the
README flags, three seeds each: F1 / bins per run.  python3 scripts/c1_hard_explore.py [variant ...]"""
import json, os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
from helpers import binning_scores, synth_sim8_c1_hard, write_fasta, C1H_GC, C1H_COVS

FLAGS = ["-k", "3", "-bc", "10", "-bs", "32", "--ae-dims", "4", "--ae-epochs", "200", "-bit", "0", "-mbs", "5000"]
WIDE = (0.36, 0.395, 0.43, 0.465, 0.50, 0.535, 0.57, 0.57)
VARIANTS = {
    "default": {},
    "div1": {"strain_div": 0.01},
    "div6": {"strain_div": 0.06},
    "nostrain": {"strain_of": {}, "gcs": (0.40, 0.42, 0.44, 0.46, 0.50, 0.52, 0.56, 0.60)},
    "ratio2": {"covs": (3100.0, 2400.0, 1900.0, 1500.0, 1200.0, 950.0, 600.0, 1200.0)},
    "w6_300_900": {"gcs": WIDE, "strain_div": 0.06, "covs": (3100.0, 2400.0, 1900.0, 1500.0, 1200.0, 950.0, 300.0, 900.0)},
    "w10_300_900": {"gcs": WIDE, "strain_div": 0.10, "covs": (3100.0, 2400.0, 1900.0, 1500.0, 1200.0, 950.0, 300.0, 900.0)},
    "c10_300_900": {"strain_div": 0.10, "covs": (3100.0, 2400.0, 1900.0, 1500.0, 1200.0, 950.0, 300.0, 900.0)},
    "w3_300_900": {"gcs": WIDE, "strain_div": 0.03, "covs": (3100.0, 2400.0, 1900.0, 1500.0, 1200.0, 950.0, 300.0, 900.0)},
    "w6_200_1000": {"gcs": WIDE, "strain_div": 0.06, "covs": (3100.0, 2400.0, 1900.0, 1500.0, 1200.0, 950.0, 200.0, 1000.0)},
    "w15_300_900": {"gcs": WIDE, "strain_div": 0.15, "covs": (3100.0, 2400.0, 1900.0, 1500.0, 1200.0, 950.0, 300.0, 900.0)},
}
out = {}
for name in (sys.argv[1:] or list(VARIANTS)):
    tmp = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None, prefix="c1h_")
    t0 = time.time()
    reads, labels = synth_sim8_c1_hard(**VARIANTS[name])
    fa = os.path.join(tmp, "reads.fasta")
    write_fasta(fa, reads)
    del reads
    tg = time.time() - t0
    res = []
    for seed in (1, 2, 3):
        o = os.path.join(tmp, f"out{seed}")
        subprocess.run([sys.executable, os.path.join(ROOT, "lrbinner.py"), "reads", "-r", fa, "-o", o] + FLAGS + ["--cuda", "-t", "32"],
                       check=True, cwd=ROOT, env=dict(os.environ, LRB_SEED=str(seed)), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        bins = [int(x) for x in open(os.path.join(o, "bins.txt")).read().split()]
        p, r, f1, nb = binning_scores(bins, labels)
        # per-genome recall of the largest bin of each genome
        res.append({"seed": seed, "precision": round(p, 3), "recall": round(r, 3), "f1": round(f1, 3), "bins": nb})
        shutil.rmtree(o)
    shutil.rmtree(tmp)
    out[name] = res
    print(name, f"gen {tg:.0f}s", res, flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04_c1_hard_explore.json"), "w"), indent=1)
