#!/usr/bin/env python3
"""The table and coverage stages of the runner path at C3's read shape (resident batches from the composition stage):
wall time of each against the sum of its kernels (run under `rocprofv3 --kernel-trace --stats` for the latter).
python3 scripts/c3_stage_probe.py [n_reads]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
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
    import torch
    from lrbinner_amd import runners_utils as ru, device as _dev
    times = {}
    if os.environ.get("C3_STAGE_CALLS"):   # wall time per library entry point
        _call = _dev.call

        def timed_call(name, *a):
            t0 = time.perf_counter()
            try:
                return _call(name, *a)
            finally:
                e = times.setdefault(name, [0, 0.0]); e[0] += 1; e[1] += time.perf_counter() - t0

        _dev.call = timed_call
    out = os.path.join(tmp, "out")
    warm = os.path.join(tmp, "warm.fasta")
    with open(warm, "wb") as f:
        f.write(b">w\n" + b"ACGT" * 100 + b"\n")
    ru.run_kmers(warm, os.path.join(tmp, "warm_out"), 4, 2)
    ru.release_resident()
    side = None
    if os.environ.get("C3_SIDE_ALLOC"):   # N segments of 17.6 GB allocated on a side thread while run_kmers runs
        import threading
        nseg = int(os.environ["C3_SIDE_ALLOC"])
        seg_t, segs = [], []

        def side_alloc():
            ctx = ru._context()
            for _ in range(nseg):
                t0 = time.perf_counter()
                segs.append(ctx.alloc(int(17.6e9)))
                seg_t.append(time.perf_counter() - t0)

        side = threading.Thread(target=side_alloc)
    for name, fn in (("run_kmers", lambda: ru.run_kmers(fa, out, 4, int(os.environ.get('C3_THREADS', '32')))),
                     ("run_15mer_counts", lambda: ru.run_15mer_counts(fa, out, int(os.environ.get('C3_THREADS', '32')), defer_table_file=True, coverage_bins=32)),
                     ("run_15mer_vecs", lambda: ru.run_15mer_vecs(fa, out, 10, 32, int(os.environ.get('C3_THREADS', '32'))))):
        torch.cuda.synchronize()
        prof = None
        if os.environ.get("C3_STAGE_PROFILE") and name != "run_kmers":
            import cProfile
            prof = cProfile.Profile(); prof.enable()
        if side is not None and name == "run_kmers":
            side.start()
        t0 = time.time(); fn(); t1 = time.time(); torch.cuda.synchronize(); t2 = time.time()
        if side is not None and name == "run_kmers":
            alive = side.is_alive()
            side.join()
            print(f"side thread: {len(seg_t)} x 17.6 GB, {sum(seg_t):.3f} s in all, still allocating when run_kmers returned: {alive}; "
                  f"ms each: {' '.join(f'{t * 1e3:.0f}' for t in seg_t)}", flush=True)
            tf = time.perf_counter()
            for p_ in segs:
                ru._context().free(p_)
            print(f"   freed again in {(time.perf_counter() - tf) * 1e3:.0f} ms", flush=True)
        print(f"{name}: {t1 - t0:.3f} s (+{t2 - t1:.3f} s until the GPU is idle)", flush=True)
        if times:
            print("   " + "; ".join(f"{k} x{v[0]} {v[1] * 1e3:.0f} ms" for k, v in sorted(times.items(), key=lambda kv: -kv[1][1])[:8]), flush=True)
            times.clear()
        if prof is not None:
            import io, pstats
            prof.disable()
            st = io.StringIO(); pstats.Stats(prof, stream=st).sort_stats("tottime").print_stats(14)
            print(st.getvalue()[:3500], flush=True)
    t0 = time.time(); ru.finish_table_files(out); print(f"table file complete after another {time.time() - t0:.3f} s")
