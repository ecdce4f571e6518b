"""Probe: run ./lrbinner.py on synthetic metagenomes of several shapes on the GPU box
and print the binning scores (used to pick the data set of the F1 gate)."""
import os, sys, subprocess, tempfile, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import synth_metagenome, write_fasta, binning_scores

configs = json.loads(sys.argv[1]) if len(sys.argv) > 1 else [dict()]
for cfg in configs:
    extra = cfg.pop("_args", [])
    reps = cfg.pop("_reps", 2)
    t0 = time.time()
    reads, labels = synth_metagenome(**cfg)
    with tempfile.TemporaryDirectory() as tmp:
        fa = os.path.join(tmp, "reads.fasta"); write_fasta(fa, reads)
        for rep in range(reps):
            out = os.path.join(tmp, "out")
            cmd = [sys.executable, os.path.join(ROOT, "lrbinner.py"), "reads", "-r", fa, "-o", out, "-k", "3",
                   "-bc", "10", "-bs", "8", "--ae-dims", "4", "--ae-epochs", "200", "-bit", "0", "-mbs", "200",
                   "--cuda", "-t", "8"] + list(map(str, extra))
            t1 = time.time()
            r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True)
            if r.returncode != 0:
                print("FAILED", cfg, r.stderr[-800:]); break
            bins = [int(x) for x in open(os.path.join(out, "bins.txt")).read().split()]
            print(json.dumps({"cfg": cfg, "args": extra, "n": len(reads), "scores": binning_scores(bins, labels),
                              "secs": round(time.time() - t1, 1)}), flush=True)
