#!/usr/bin/env python3
"""Contigs FASTA parse: lrb_fasta_scan against the Python line loop, unwrapped and wrapped at 60 columns
(python scripts/contigs_parse_probe.py [n_contigs] [length])."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lrbinner_amd import runners_utils as ru

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
rng = np.random.default_rng(0)
seq = bytes(rng.choice(list(b"ACGT"), L).astype(np.uint8))
for wrap in (10 ** 9, 60, 10 ** 9, 60):
    p = f"/dev/shm/contigs_probe_{wrap}.fasta"
    with open(p, "wb") as f:
        for i in range(n):
            f.write(b">c%d\n" % i)
            for a in range(0, L, wrap):
                f.write(seq[a:a + wrap] + b"\n")
    t = time.time(); cf = ru._NativeContigs(p); t_native = time.time() - t
    t = time.time(); total = sum(len(cf[i]) for i in range(cf.n)); t_bytes = time.time() - t
    cf.close()
    t = time.time(); keep = list(ru._fasta_records_b(p)); t_py = time.time() - t
    assert total == n * L == sum(len(s) for _, s in keep)
    print(f"{os.path.getsize(p) / 1e9:.2f} GB, lines of {'one record' if wrap > L else wrap}: native scan {t_native:.2f} s "
          f"(+ {t_bytes:.2f} s to hand every record out as bytes), Python line loop {t_py:.2f} s", flush=True)
    del keep
    os.remove(p)
