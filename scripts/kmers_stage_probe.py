#!/usr/bin/env python3
"""Where the composition stage (run_kmers, file -> com_profs) spends its wall time at C3's read shape: the stage as it
is, with the writer thread's two file writes turned into no-ops (what parse + pack + upload + K1 + K8 + download cost
alone), and the raw rate of one thread writing the same bytes to tmpfs.
python scripts/kmers_stage_probe.py [n_reads]"""
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
    rows = np.empty((block, L + 1), dtype=np.uint8)
    rows[:, :L] = letters[rng.integers(0, 4, size=(block, L), dtype=np.uint8)]; rows[:, L] = 10
    with open(fa, "wb") as f:
        for s in range(0, n, block):
            rows[:, :L] = np.roll(rows[:, :L], 37, axis=1)
            for i in range(min(block, n - s)):
                f.write(b">r%d\n" % (s + i)); f.write(rows[i].tobytes())
    gb = os.path.getsize(fa) / 1e9
    from lrbinner_amd import runners_utils as ru
    out = os.path.join(tmp, "out")
    warm = os.path.join(tmp, "warm.fasta")
    with open(warm, "wb") as f:
        f.write(b">w\n" + b"ACGT" * 100 + b"\n")
    ru.run_kmers(warm, os.path.join(tmp, "warm_out"), 4, 2)
    ru.release_resident()
    for threads in (32, 64):
        t0 = time.time(); ru.run_kmers(fa, out, 4, threads); dt = time.time() - t0
        ru.release_resident()
        print(f"run_kmers k=4, {n} reads ({gb:.1f} GB), {threads} reader threads: {dt:.2f} s = {gb / dt:.1f} GB/s of FASTA", flush=True)
    text_bytes = os.path.getsize(f"{out}/profiles/com_profs"); side_bytes = os.path.getsize(f"{out}/profiles/com_profs.q6")
    # the writer's writes dropped
    orig = ru._ProfileWriter._run

    def dropping(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            self.free[item[0]].release()

    ru._ProfileWriter._run = dropping
    t0 = time.time(); ru.run_kmers(fa, out, 4, 32); dt = time.time() - t0
    ru.release_resident()
    ru._ProfileWriter._run = orig
    print(f"the same with the writer thread dropping what it gets: {dt:.2f} s", flush=True)
    # one thread writing text_bytes + side_bytes to tmpfs in 64 MB pieces
    buf = bytes(64 << 20)
    t0 = time.time()
    with open(os.path.join(tmp, "w.bin"), "wb") as f:
        left = text_bytes + side_bytes
        while left > 0:
            f.write(buf[: min(left, len(buf))]); left -= len(buf)
    dt = time.time() - t0
    print(f"one thread writing {(text_bytes + side_bytes) / 1e9:.2f} GB (text {text_bytes / 1e9:.2f} + side-car {side_bytes / 1e9:.2f}) to tmpfs: {dt:.2f} s = {(text_bytes + side_bytes) / 1e9 / dt:.1f} GB/s")
