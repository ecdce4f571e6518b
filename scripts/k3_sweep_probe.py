#!/usr/bin/env python3
"""K3 as a sweep over the map (lrb_cov_hist_sweep_dev) against the gather form (lrb_cov_hist_map_dev): same
histograms, timing by size.  python scripts/k3_sweep_probe.py [n_reads ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from lrbinner_amd import device as lrb
import bench

ctx = lrb.Context(0, use_torch_stream=True)
dev = torch.device("cuda")
L = 10_000
sizes = [int(a) for a in sys.argv[1:]] or [50_000, 400_000]
table = torch.zeros(lrb.K15_ENTRIES, dtype=torch.int32, device=dev)

def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

for n in sizes:
    codes, mask, co, mo, lens, words = bench.synth_packed(torch, n, L, 5, dev)
    pr = lrb.PackedReads(codes, mask, co, mo, lens, n)
    table.zero_()
    ctx.k15_accumulate_part_dev(pr, table, n * L)
    sub = lrb.PackedReads(codes, mask, co[: n // 4 + 1].contiguous(), mo[: n // 4 + 1].contiguous(), lens[: n // 4].contiguous(), n // 4)
    for _ in range(6):
        ctx.k15_accumulate_part_dev(sub, table, (n // 4) * L)
    ctx.k15_mirror_dev(table)
    for bs, bins in ((10, 32), (2, 10), (1, 256)):
        m = ctx.cov_map_build_dev(table, bs, bins)
        h0, s0 = ctx.cov_hist_map_dev(pr, m, bins)
        h1, s1 = ctx.cov_hist_sweep_dev(pr, m, bins)
        torch.cuda.synchronize()
        eq = bool(torch.equal(h0, h1)), bool(torch.equal(s0, s1))
        h = torch.empty((n, bins), dtype=torch.int32, device=dev); s = torch.empty(n, dtype=torch.int32, device=dev)
        t_g = timed(lambda: ctx.cov_hist_map_dev(pr, m, bins, hist=h, sums=s))
        t_s = timed(lambda: ctx.cov_hist_sweep_dev(pr, m, bins, hist=h, sums=s))
        print(f"n={n} bs={bs} bins={bins}: equal {eq}  gather {t_g:.2f} ms = {n / t_g / 1e3:.2f} M reads/s   "
              f"sweep {t_s:.2f} ms = {n / t_s / 1e3:.2f} M reads/s", flush=True)
        del m, h0, s0, h1, s1, h, s
    del codes, mask, co, mo, lens, pr, sub
    torch.cuda.empty_cache()
