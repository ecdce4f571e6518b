#!/usr/bin/env python3
"""run_15mer_vecs with K3 as per-batch gathers (LRB_K3_SWEEP=0) and as a sweep over groups of resident batches:
wall time on n synthetic 10 kb reads in tmpfs.  python scripts/vecs_probe.py [n_reads]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
L = 10_000
rng = np.random.default_rng(1)
with tempfile.TemporaryDirectory(dir="/dev/shm") as tmp:
    fa = os.path.join(tmp, "reads.fasta")
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    block = 20000
    rows = np.empty((block, L + 1), dtype=np.uint8); rows[:, L] = 10
    rows[:, :L] = letters[rng.integers(0, 4, size=(block, L), dtype=np.uint8)]
    with open(fa, "wb") as f:
        for s in range(0, n, block):
            m = min(block, n - s)
            rows[: block // 10, :L] = letters[rng.integers(0, 4, size=(block // 10, L), dtype=np.uint8)]
            rows[:, :L] = np.roll(rows[:, :L], 37, axis=1)
            rows[:] = np.roll(rows, 1, axis=0)
            for i in range(m):
                f.write(b">r%d\n" % (s + i)); f.write(rows[i].tobytes())
    from lrbinner_amd import runners_utils as ru
    out = os.path.join(tmp, "out")
    ru.run_15mer_counts(fa, out, 32)
    import hashlib
    for mode in ("0", "1", "0", "1"):
        os.environ["LRB_K3_SWEEP"] = mode
        t0 = time.time()
        ru_key = os.path.abspath(out)
        # keep the table and the resident reads across repetitions: re-arm what run_15mer_vecs releases
        tab = ru._table_cache.get(ru_key)
        res = dict(ru._resident)
        orig_drop, orig_rel = ru._drop_table, ru.release_resident
        ru._drop_table = lambda *_a, **_k: None
        ru.release_resident = lambda *_a, **_k: None
        try:
            ru.run_15mer_vecs(fa, out, 10, 32, 32)
        finally:
            ru._drop_table, ru.release_resident = orig_drop, orig_rel
        dt = time.time() - t0
        h = hashlib.md5(open(os.path.join(out, "profiles/cov_profs"), "rb").read()).hexdigest()
        print(f"LRB_K3_SWEEP={mode}: {dt:.2f} s = {n / dt / 1e6:.2f} M reads/s  md5 {h}", flush=True)
