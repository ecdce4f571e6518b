#!/usr/bin/env python3
"""End-to-end (file in -> profile files out) timing of the runner shims on one GPU:
FASTA parse + PCIe + kernels + text formatting.  python scripts/e2e_scale.py [n_reads]"""
import json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from lrbinner_amd import runners_utils as ru, pipelines

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
L = 10_000
rng = np.random.default_rng(1)
res = {"n_reads": n, "read_len": L}
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
    fa = os.path.join(tmp, "reads.fasta")
    t0 = time.time()
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    with open(fa, "wb") as f:
        for s in range(0, n, 20000):
            m = min(20000, n - s)
            seqs = letters[rng.integers(0, 4, size=(m, L), dtype=np.uint8)]
            rows = np.empty((m, L + 1), dtype=np.uint8); rows[:, :L] = seqs; rows[:, L] = 10
            for i in range(m):
                f.write(b">r%d\n" % (s + i)); f.write(rows[i].tobytes())
    res["fasta_GB"] = os.path.getsize(fa) / 1e9
    res["gen_s"] = time.time() - t0
    out = os.path.join(tmp, "out")
    # one tiny call first: HIP context creation and library load are not per-file costs
    warm = os.path.join(tmp, "warm.fasta")
    with open(warm, "wb") as f:
        f.write(b">w\n" + b"ACGT" * 100 + b"\n")
    t0 = time.time(); ru.run_kmers(warm, os.path.join(tmp, "warm_out"), 3, 2); res["init_s"] = round(time.time() - t0, 3)
    ru.release_resident()
    for name, fn in (("run_kmers_k3", lambda: ru.run_kmers(fa, out, 3, 16)),
                     ("run_kmers_k4", lambda: ru.run_kmers(fa, out, 4, 16)),
                     ("run_15mer_counts", lambda: ru.run_15mer_counts(fa, out, 16)),
                     ("run_15mer_vecs", lambda: ru.run_15mer_vecs(fa, out, 10, 32, 16)),
                     ("text_to_npy", lambda: pipelines._profiles_to_npy(out))):
        t0 = time.time(); fn(); dt = time.time() - t0
        res[name] = {"s": round(dt, 3), "reads_per_s": round(n / dt)}
print(json.dumps(res, indent=1))
