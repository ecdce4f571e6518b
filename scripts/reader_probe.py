import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lrbinner_amd import device
n, L = 200000, 10000
rng = np.random.default_rng(1)
with tempfile.TemporaryDirectory(dir="/dev/shm") as tmp:
    fa = os.path.join(tmp, "r.fa")
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    with open(fa, "wb") as f:
        for s in range(0, n, 20000):
            seqs = letters[rng.integers(0, 4, size=(20000, L), dtype=np.uint8)]
            rows = np.empty((20000, L + 1), dtype=np.uint8); rows[:, :L] = seqs; rows[:, L] = 10
            for i in range(20000):
                f.write(b">r%d\n" % (s + i)); f.write(rows[i].tobytes())
    gb = os.path.getsize(fa) / 1e9
    chunks = [int(a) for a in sys.argv[1:]] or [1 << 28]
    for chunk in chunks:
      for thr in (1, 4, 8, 16, 32):
        t0 = time.time(); tot = 0
        with device.ParallelReader(fa, threads=thr, chunk_bytes=chunk) as rd:
            while True:
                b = rd.next_batch(copy=False)
                if b is None: break
                tot += len(b[1]) - 1
        dt = time.time() - t0
        print(f"parallel chunk={chunk>>20} MB threads={thr:2d}: {dt:.3f}s {gb/dt:.2f} GB/s reads={tot}", flush=True)
    t0 = time.time(); tot = 0
    with device.FastxReader(fa) as rd:
        while True:
            b = rd.next_batch(1 << 17, 1 << 29)
            if b is None: break
            tot += len(b[1]) - 1
    dt = time.time() - t0
    print(f"serial (+numpy copy): {dt:.3f}s {gb/dt:.2f} GB/s reads={tot}")
    ctx = device.Context(0)
    for thr in (8, 16):
        t0 = time.time(); tp = 0.0
        with device.ParallelReader(fa, threads=thr, chunk_bytes=1 << 28) as rd:
            while True:
                b = rd.next_batch(copy=False)
                if b is None: break
                t1 = time.time(); pb = ctx.packed_create(b[0], b[1], with_planes=True); c = pb.kmer_counts(3); pb.free(); tp += time.time() - t1
        print(f"read+upload+K1 threads={thr}: total {time.time()-t0:.3f}s of which gpu-side calls {tp:.3f}s")
