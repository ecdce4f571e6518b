#!/usr/bin/env python3
"""K3: gathers from the 4 GiB table against gathers from the compact map (512 MB of bin ids of the
canonical half).  Same histograms; timing at 200 k x 10 kb.  python scripts/k3_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from lrbinner_amd import device as lrb
import bench

ctx = lrb.Context(0, use_torch_stream=True)
dev = torch.device("cuda")
n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000, 10_000
codes, mask, co, mo, lens, words = bench.synth_packed(torch, n, L, 5, dev)
pr = lrb.PackedReads(codes, mask, co, mo, lens, n)
table = torch.zeros(lrb.K15_ENTRIES, dtype=torch.int32, device=dev)
ctx.k15_accumulate_part_dev(pr, table, n * L)
# a second, skewed layer so that the counts spread over many bins
sub = lrb.PackedReads(codes, mask, co[: n // 4 + 1].contiguous(), mo[: n // 4 + 1].contiguous(), lens[: n // 4].contiguous(), n // 4)
for _ in range(6):
    ctx.k15_accumulate_part_dev(sub, table, (n // 4) * L)
ctx.k15_mirror_dev(table)
torch.cuda.synchronize()

def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

for bs, bins in ((10, 32), (2, 10), (4, 10), (1, 256)):
    h0, s0 = ctx.cov_hist_dev(pr, table, bs, bins)
    t_build = timed(lambda: ctx.cov_map_build_dev(table, bs, bins))
    m = ctx.cov_map_build_dev(table, bs, bins)
    h1, s1 = ctx.cov_hist_map_dev(pr, m, bins)
    torch.cuda.synchronize()
    print(f"bs={bs} bins={bins}: equal hist {bool(torch.equal(h0, h1))} sums {bool(torch.equal(s0, s1))} "
          f"nonzero bins {int((h0.sum(0) > 0).sum())}  map build {t_build:.2f} ms")
h = torch.empty((n, 32), dtype=torch.int32, device=dev); s = torch.empty(n, dtype=torch.int32, device=dev)
t_old = timed(lambda: ctx.cov_hist_dev(pr, table, 10, 32, hist=h, sums=s))
m = ctx.cov_map_build_dev(table, 10, 32)
t_new = timed(lambda: ctx.cov_hist_map_dev(pr, m, 32, hist=h, sums=s))
print(f"K3 table: {t_old:.2f} ms = {n / t_old / 1e3:.2f} M reads/s   map: {t_new:.2f} ms = {n / t_new / 1e3:.2f} M reads/s")
