#!/usr/bin/env python3
"""The parser pool alone on a 20 GB FASTA of 10 kb reads in tmpfs: batches taken and dropped, GB/s by thread count
(chunk 64 MB as the runners use), handing out ASCII batches and batches packed on the host (2 bits a base + mask, round 5).
python3 scripts/parser_probe.py [n_reads]"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lrbinner_amd import device
n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000, 10000
rng = np.random.default_rng(1)
with tempfile.TemporaryDirectory(dir="/dev/shm") as tmp:
    fa = os.path.join(tmp, "r.fa")
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    rows = np.empty((20000, L + 1), dtype=np.uint8)
    rows[:, :L] = letters[rng.integers(0, 4, size=(20000, L), dtype=np.uint8)]; rows[:, L] = 10
    with open(fa, "wb") as f:
        for s in range(0, n, 20000):
            rows[:, :L] = np.roll(rows[:, :L], 37, axis=1)
            for i in range(min(20000, n - s)):
                f.write(b">r%d\n" % (s + i)); f.write(rows[i].tobytes())
    gb = os.path.getsize(fa) / 1e9
    for rep in range(2):
        for packed in (False, True):
            for thr in (8, 16, 32, 64, 128):
                t0 = time.time(); tot = 0
                with device.ParallelReader(fa, threads=thr, chunk_bytes=1 << 26, packed=packed) as rd:
                    while True:
                        if packed:
                            b = rd.next_packed()
                            if b is None: break
                            tot += b.n
                        else:
                            b = rd.next_batch(copy=False)
                            if b is None: break
                            tot += len(b[1]) - 1
                dt = time.time() - t0
                print(f"parser pool alone, {'packed' if packed else 'ASCII '} batches, {thr:3d} threads: {dt:.3f} s = {gb / dt:.1f} GB/s ({tot} reads)", flush=True)
