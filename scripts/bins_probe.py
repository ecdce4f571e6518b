#!/usr/bin/env python3
"""Wall time of the three drop-in executables (lrbinner_amd/bin) on n synthetic 10 kb reads in tmpfs, next to
the reference's own binaries (oracle/_ref) on a sample of the same file.  python scripts/bins_probe.py [n]"""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import numpy as np

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
L = 10_000
rng = np.random.default_rng(2)
letters = np.frombuffer(b"ACGT", dtype=np.uint8)
res = {"n_reads": n, "read_len": L, "threads": os.cpu_count()}
with tempfile.TemporaryDirectory(dir="/dev/shm") as tmp:
    fa, small = os.path.join(tmp, "reads.fasta"), os.path.join(tmp, "sample.fasta")
    block = 20000
    rows = np.empty((block, L + 1), dtype=np.uint8)
    rows[:, :L] = letters[rng.integers(0, 4, size=(block, L), dtype=np.uint8)]
    rows[:, L] = 10
    with open(fa, "wb") as f, open(small, "wb") as g:
        for s in range(0, n, block):
            rows[:, :L] = np.roll(rows[:, :L], 37, axis=1)
            rows[:] = np.roll(rows, 1, axis=0)
            for i in range(min(block, n - s)):
                rec = b">r%d\n" % (s + i) + rows[i].tobytes()
                f.write(rec)
                if s + i < 100_000:
                    g.write(rec)
    t = str(os.cpu_count())

    def timed(cmd):
        t0 = time.time()
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)
        return time.time() - t0

    B = os.path.join(ROOT, "lrbinner_amd", "bin")
    timed([f"{B}/count-kmers", small, f"{tmp}/w", "3", t])   # first touch of the device and the code objects
    for k in (3, 4):
        dt = timed([f"{B}/count-kmers", fa, f"{tmp}/com{k}", str(k), t])
        res[f"count-kmers k={k}"] = {"s": round(dt, 2), "reads_per_s": round(n / dt)}
    dt = timed([f"{B}/count-15mers", fa, f"{tmp}/table", t])
    res["count-15mers"] = {"s": round(dt, 2), "reads_per_s": round(n / dt)}
    dt = timed([f"{B}/search-15mers", f"{tmp}/table", fa, f"{tmp}/cov", "10", "32", t])
    res["search-15mers"] = {"s": round(dt, 2), "reads_per_s": round(n / dt)}
    R = os.path.join(ROOT, "oracle", "_ref")
    if os.path.exists(f"{R}/count-kmers"):
        m = 100_000
        dt = timed([f"{R}/count-kmers", small, f"{tmp}/rcom", "3", t])
        res["reference count-kmers k=3 (100 k reads)"] = {"s": round(dt, 2), "reads_per_s": round(m / dt)}
        same = open(f"{tmp}/rcom", "rb").read() == open(f"{tmp}/com3", "rb").read()[: os.path.getsize(f"{tmp}/rcom")]
        res["first 100 k rows of com_profs equal the reference's"] = bool(same)
        dt = timed([f"{R}/count-15mers", small, f"{tmp}/rtable", t])
        res["reference count-15mers (100 k reads)"] = {"s": round(dt, 2), "reads_per_s": round(m / dt)}
        dt = timed([f"{R}/search-15mers", f"{tmp}/rtable", small, f"{tmp}/rcov", "10", "32", t])
        res["reference search-15mers (100 k reads)"] = {"s": round(dt, 2), "reads_per_s": round(m / dt)}
print(json.dumps(res, indent=1))
