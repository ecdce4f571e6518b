#!/usr/bin/env python3
"""Per-batch cost of the resident path at C3's read shape, the file out of the picture: packed_create (upload + pack +
the transposed layout of the k given) and kmer_text (K1 + K8 + download) on batches of 6,600 reads of 10 kb that are
already parsed.  python3 scripts/packed_create_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from lrbinner_amd import device as lrb

n, L, reps = 6600, 10_000, 60
rng = np.random.default_rng(1)
seqs = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n * L, dtype=np.uint8)]
offs = (np.arange(n + 1, dtype=np.uint64) * L)
ctx = lrb.Context(0)
for planes, k in ((2, 4), (1, 3)):
    bs = [ctx.packed_create(seqs, offs, with_planes=planes) for _ in range(3)]
    t0 = time.perf_counter()
    bs += [ctx.packed_create(seqs, offs, with_planes=planes) for _ in range(reps)]
    t1 = time.perf_counter()
    for b in bs[:3]:
        b.kmer_text(k)
    t2 = time.perf_counter()
    for b in bs[3:]:
        b.kmer_text(k)
    t3 = time.perf_counter()
    for b in bs:
        b.free()
    t4 = time.perf_counter()
    mb = n * L / 1e6
    print(f"k = {k}: packed_create {1e3 * (t1 - t0) / reps:.2f} ms per batch of {mb:.0f} MB ({mb / 1e3 / ((t1 - t0) / reps):.1f} GB/s), "
          f"kmer_text {1e3 * (t3 - t2) / reps:.2f} ms, free {1e3 * (t4 - t3) / (reps + 3):.2f} ms", flush=True)
