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
read_len = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
rng = np.random.default_rng(8)
# eight genomes with their own base composition and abundance; reads = windows with 10 % substitutions
n_genomes, glen, err = 8, 1_500_000, 0.10
cov = np.array([4, 6, 9, 13, 19, 28, 41, 60], dtype=np.float64)
share = cov / cov.sum()
letters = np.frombuffer(b"ACGT", dtype=np.uint8)
genomes = []
for g in range(n_genomes):
    p = rng.dirichlet(np.full(4, 6.0))
    # order-1 structure: mix of two compositions along the genome in 5 kb blocks
    q = rng.dirichlet(np.full(4, 6.0))
    blocks = rng.random(glen // 5000 + 1) < 0.5
    prob = np.where(np.repeat(blocks, 5000)[:glen, None], p[None, :], q[None, :])
    u = rng.random(glen)
    genomes.append(letters[(u[:, None] > np.cumsum(prob, axis=1)).sum(1).clip(0, 3)])
origin = rng.choice(n_genomes, size=n_reads, p=share)
t0 = time.time()
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
    fa = os.path.join(tmp, "reads.fasta")
    with open(fa, "wb") as f:
        for s in range(0, n_reads, 20000):
            m = min(20000, n_reads - s)
            starts = rng.integers(0, glen - read_len, size=m)
            rows = np.empty((m, read_len + 1), dtype=np.uint8)
            for i in range(m):
                rows[i, :read_len] = genomes[origin[s + i]][starts[i]:starts[i] + read_len]
            sub = rng.random((m, read_len)) < err
            rows[:, :read_len][sub] = letters[rng.integers(0, 4, size=int(sub.sum()))]
            rows[:, read_len] = 10
            for i in range(m):
                f.write(b">r%d\n" % (s + i)); f.write(rows[i].tobytes())
    gen_s = time.time() - t0
    out = os.path.join(tmp, "out")
    cmd = [sys.executable, os.path.join(ROOT, "lrbinner.py"), "reads", "-r", fa, "-o", out, "-k", "3", "-bc", "10",
           "-bs", "32", "--ae-dims", "4", "--ae-epochs", "200", "-bit", "0", "-mbs", "1000", "--cuda", "-t", "16"]
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
