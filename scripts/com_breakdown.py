#!/usr/bin/env python3
"""Where run_kmers spends its time at C3 shape (k = 4, 10 kb reads): parse+pack, K1 (+D2H), text
formatting, file writes.  python scripts/com_breakdown.py [n_reads] [threads]"""
import json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 32
L = 10_000
rng = np.random.default_rng(1)
res = {"n_reads": n, "threads": threads, "host_cpus": os.cpu_count()}
with tempfile.TemporaryDirectory(dir="/dev/shm") as tmp:
    fa = os.path.join(tmp, "reads.fasta")
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    block = 20000
    rows = np.empty((block, L + 1), dtype=np.uint8)
    rows[:, :L] = letters[rng.integers(0, 4, size=(block, L), dtype=np.uint8)]
    rows[:, L] = 10
    with open(fa, "wb") as f:
        for s in range(0, n, block):
            m = min(block, n - s)
            rows[:, :L] = np.roll(rows[:, :L], 37, axis=1)
            for i in range(m):
                f.write(b">r%d\n" % (s + i)); f.write(rows[i].tobytes())
    from lrbinner_amd import runners_utils as ru, device as lrb
    warm = os.path.join(tmp, "warm.fasta")
    with open(warm, "wb") as f:
        f.write(b">w\n" + b"ACGT" * 100 + b"\n")
    ru.run_kmers(warm, os.path.join(tmp, "warm_out"), 4, 2)
    ru.release_resident()
    for mode in ("device_text", "host_text"):
        T = {"parse_pack": 0.0, "k1_format_d2h": 0.0, "write_text": 0.0, "write_sidecar": 0.0}
        os.makedirs(os.path.join(tmp, mode, "profiles"))
        out_path = os.path.join(tmp, mode, "profiles/com_profs")
        ru.release_resident()
        t_all = time.time()
        with open(out_path, "wb") as out:
            side = ru._ValueSidecar(out_path)
            it = ru._resident_batches(fa, with_planes=False, threads=threads)
            while True:
                t0 = time.time()
                try:
                    batch = next(it)
                except StopIteration:
                    T["reader_close"] = time.time() - t0
                    break
                T["parse_pack"] += time.time() - t0
                t0 = time.time()
                if mode == "device_text":   # K1 + K8 in HBM, text and six-decimal integers come back
                    txt, q = batch.kmer_text(4)
                else:                       # K1 in HBM, counts come back, 32 host threads format them
                    counts = batch.kmer_counts(4)
                    txt, vals = lrb.format_com(counts, batch.lens, 4, threads=threads, want_values=True)
                    q = np.rint(vals * 1e6).astype(np.uint32)
                T["k1_format_d2h"] += time.time() - t0
                t0 = time.time(); out.write(txt); T["write_text"] += time.time() - t0
                t0 = time.time(); side.append(q); T["write_sidecar"] += time.time() - t0
            side.close()
        res[mode] = {"total_s": round(time.time() - t_all, 3), "stages_s": {k: round(v, 3) for k, v in T.items()}}
    a = open(os.path.join(tmp, "device_text/profiles/com_profs"), "rb").read()
    b = open(os.path.join(tmp, "host_text/profiles/com_profs"), "rb").read()
    res["same_bytes"] = a == b
    res["text_GB"] = round(len(a) / 1e9, 3)
    # ---- the 15-mer table stage and the coverage stage on the batches now resident in HBM ----
    ctx = ru._context()
    T = {}
    t_all = time.time()
    t0 = time.time(); table = ctx.alloc_table(); T["alloc_zero_table"] = time.time() - t0
    t0 = time.time()
    for batch in ru._resident_batches(fa, threads=threads):
        batch.k15_accumulate(table)
    ctx.sync(); T["k2_accumulate_batch_by_batch"] = time.time() - t0
    ctx.memset(table, 0, 4 * lrb.K15_ENTRIES); ctx.sync()
    t0 = time.time()
    ctx.k15_accumulate_many(list(ru._resident_batches(fa, threads=threads)), table)
    ctx.sync(); T["k2_accumulate_grouped"] = time.time() - t0
    t0 = time.time(); ctx.k15_mirror(table); ctx.sync(); T["mirror"] = time.time() - t0
    t0 = time.time(); ctx.k15_write_file(table, os.path.join(tmp, "table")); T["write_table_file"] = time.time() - t0
    res["table_stage"] = {"total_s": round(time.time() - t_all, 3), "stages_s": {k: round(v, 3) for k, v in T.items()}}
    T = {"k3_format_d2h": 0.0, "write_text": 0.0, "write_sidecar": 0.0}
    cov_path = os.path.join(tmp, "cov_profs")
    t_all = time.time()
    with open(cov_path, "wb") as out:
        side = ru._ValueSidecar(cov_path)
        for batch in ru._resident_batches(fa, threads=threads):
            t0 = time.time(); txt, q = batch.cov_text(table, 10, 32); T["k3_format_d2h"] += time.time() - t0
            t0 = time.time(); out.write(txt); T["write_text"] += time.time() - t0
            t0 = time.time(); side.append(q); T["write_sidecar"] += time.time() - t0
        side.close()
    res["coverage_stage"] = {"total_s": round(time.time() - t_all, 3), "stages_s": {k: round(v, 3) for k, v in T.items()}}
    ctx.free(table)
    ru.release_resident()
print(json.dumps(res, indent=1))
