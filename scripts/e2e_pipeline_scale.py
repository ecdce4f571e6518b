#!/usr/bin/env python3
"""Whole `lrbinner.py reads` run at Sim-8 scale on one GPU: per-stage wall times from the log,
reads binned per second end to end, and the binning scores against the known origin of every
read.  python scripts/e2e_pipeline_scale.py [n_reads] [read_len] > gpurun_out/e2e_pipeline.json"""
import json, os, re, subprocess, sys, tempfile, time
from datetime import datetime
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import binning_scores

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 432_333
# the 8-genome stand-in of the accuracy gate (tests/helpers.synth_sim8: order-3 Markov genomes, 5x-60x,
# 10 kb reads with ~10 % noise), genome lengths scaled so that it yields n_reads reads.  Round 1 used a
# block-mixture generator here whose genomes each carry two base compositions -- the 14-bin / recall-55
# result in profiles/r01_e2e_pipeline.json is that data's, see DESIGN.md section 5.
from helpers import synth_sim8, write_fasta
t0 = time.time()
reads, origin = synth_sim8(scale=n_reads / 40350.0)
n_reads, read_len = len(reads), 10_000
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
    fa = os.path.join(tmp, "reads.fasta")
    write_fasta(fa, reads)
    del reads
    gen_s = time.time() - t0
    out = os.path.join(tmp, "out")
    cmd = [sys.executable, os.path.join(ROOT, "lrbinner.py"), "reads", "-r", fa, "-o", out, "-k", "3", "-bc", "10",
           "-bs", "2", "--ae-dims", "4", "--ae-epochs", "200", "-bit", "0", "-mbs", "5000", "--cuda", "-t", "16"]
    t1 = time.time()
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True)
    wall = time.time() - t1
    if r.returncode != 0:
        print(r.stderr[-3000:]); sys.exit(1)
    stamps = []
    for line in open(os.path.join(out, "LRBinner.log")):
        m = re.match(r"(\d{4}-\d\d-\d\d \d\d:\d\d:\d\d,\d{3}) - \w+ - (.*)", line)
        if m:
            stamps.append((datetime.strptime(m.group(1), "%Y-%m-%d %H:%M:%S,%f"), m.group(2).strip()))
    stages = [{"message": b[1], "seconds_since_previous": round((b[0] - a[0]).total_seconds(), 3)}
              for a, b in zip(stamps, stamps[1:]) if (b[0] - a[0]).total_seconds() >= 0.05]
    def at(msg):
        for t, m in stamps:
            if m.startswith(msg):
                return t
        return None
    spans = {}
    for name, a, b in (("profiles (composition, 15-mer table, coverage)", "Counting", "Computing 15-mer profiles complete"),
                       ("text -> npy + VAE training + encode", "Computing 15-mer profiles complete", "VAE training complete"),
                       ("clustering + left-over assignment + output", "VAE training complete", "Program Finished")):
        ta, tb = at(a), at(b)
        if ta and tb:
            spans[name] = round((tb - ta).total_seconds(), 3)
    bins = [int(x) for x in open(os.path.join(out, "bins.txt")).read().split()]
    res = {"n_reads": n_reads, "read_len": read_len, "fasta_GB": round(os.path.getsize(fa) / 1e9, 3),
           "command": " ".join(cmd[1:]).replace(tmp, "$TMP"), "wall_s": round(wall, 2),
           "reads_binned_per_s_end_to_end": round(n_reads / wall), "stage_seconds": spans,
           "log_gaps": [g for g in stages if not g["message"].startswith("Epoch")],
           "scores": binning_scores(bins, origin.tolist()), "generate_s": round(gen_s, 1)}
    print(json.dumps(res, indent=1))
