#!/usr/bin/env python3
"""k=4 lane-per-read kernel (lrb_kmer_counts4t_dev) against the oracle and the LDS kernel, then timing
at 1 M x 10 kb.  python scripts/k4_probe.py > gpurun_out/k4_probe.txt"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from helpers import random_reads
from oracle import oracle as orc
from lrbinner_amd import device as lrb
import bench

ctx = lrb.Context(0, use_torch_stream=True)
rng = np.random.default_rng(3)
sets = {
    "ragged": random_reads(rng, 1000, 0, 3000, p_n=0.01, p_lower=0.01),
    "short": [b"", b"A", b"ACG", b"ACGT", b"ACGTA", b"N" * 70, b"acgtnACGT" * 9] + random_reads(rng, 150, 60, 70),
    "long": [b"A" * 200_000, bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 150_000)), b"ACGT" * 20_000] + random_reads(rng, 70, 100, 5000),
    "equal": random_reads(rng, 500, 10_000, 10_000),
}
for name, reads in sets.items():
  for k in (4, 5):
    buf, offs = orc.concat(reads)
    want, _ = orc.count_kmers(buf, offs, k)
    for sort in (True, False):
        pr = ctx.pack(torch.from_numpy(buf).cuda(), offs, want_mask=False)
        ctx.make_codes_t(pr, sort=sort)
        got = ctx.kmer_counts4t_dev(pr, k=k).cpu().numpy().view(np.uint32)
        ok = np.array_equal(got, want)
        print(name, "k", k, "sort" if sort else "as given", len(reads), "reads:", "OK" if ok else "MISMATCH", flush=True)
        if not ok:
            bad = np.nonzero((got != want).any(1))[0]
            print("  rows", bad[:10], [len(reads[i]) for i in bad[:10]])
            i = bad[0]
            print("  got ", got[i][:16], got[i].sum(), "\n  want", want[i][:16], want[i].sum())

n, L = 1_000_000, 10_000
codes, mask, co, mo, lens, words = bench.synth_packed(torch, n, L, 12345, torch.device("cuda"))
pr = lrb.PackedReads(codes, mask, co, mo, lens, n)
t0 = time.perf_counter(); ctx.make_codes_t(pr, sort=True); torch.cuda.synchronize()
print("make_codes_t %.1f ms" % ((time.perf_counter() - t0) * 1e3))
for k in (4, 5):
  dim = lrb.kmer_dim(k)
  out_a = torch.empty((n, dim), dtype=torch.int32, device="cuda")
  out_b = torch.empty((n, dim), dtype=torch.int32, device="cuda")
  for name, fn in ((f"lane k={k}", lambda: ctx.kmer_counts4t_dev(pr, out=out_a, k=k)), (f"lds k={k}", lambda: ctx.kmer_counts_dev(pr, k, out=out_b))):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    print(f"{name}: {ms:.3f} ms  {n / ms / 1e3:.1f} M reads/s  roofline {(2500 + 4 * dim) * n / (ms * 1e-3) / 8e12:.3f}")
  print("1M equal:", bool(torch.equal(out_a, out_b)))
