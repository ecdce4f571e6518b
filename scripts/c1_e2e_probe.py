#!/usr/bin/env python3
"""`lrbinner.py reads` with the README's flags on the C1 stand-in (tests/helpers.synth_sim8_c1), a few runs: wall time
of the process and the stage stamps of its log.
python3 scripts/c1_e2e_probe.py [runs]"""
import os, re, subprocess, sys, tempfile, time
from datetime import datetime
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import synth_sim8_c1, write_fasta

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
flags = "-k 3 -bc 10 -bs 32 --ae-dims 4 --ae-epochs 200 -bit 0 -mbs 5000".split()
with tempfile.TemporaryDirectory(dir="/dev/shm") as tmp:
    reads, labels = synth_sim8_c1()
    fa = os.path.join(tmp, "reads.fasta")
    write_fasta(fa, reads)
    del reads
    for early in ["-"] * runs:
        o = os.path.join(tmp, "out")
        cmd = [sys.executable, os.path.join(ROOT, "lrbinner.py"), "reads", "-r", fa, "-o", o] + flags + ["--cuda", "-t", "32"]
        t0 = time.time()
        subprocess.run(cmd, check=True, cwd=ROOT, env=dict(os.environ, LRB_SEED="1"),
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        wall = time.time() - t0
        stamps = []
        for line in open(os.path.join(o, "LRBinner.log")):
            m = re.match(r"(\d{4}-\d\d-\d\d \d\d:\d\d:\d\d,\d{3}) - \w+ - (.*)", line)
            if m and not m.group(2).startswith("Epoch"):
                stamps.append((datetime.strptime(m.group(1), "%Y-%m-%d %H:%M:%S,%f"), m.group(2).strip()))
        first = stamps[0][0]
        marks = [(round((t - first).total_seconds(), 2), msg[:34]) for t, msg in stamps
                 if any(k in msg for k in ("Command", "complete", "detected", "Finished", "VAE training information"))]
        print(f"wall {wall:.2f} s, log starts {wall - (stamps[-1][0] - first).total_seconds():.2f} s after the process; {marks}", flush=True)
