#!/usr/bin/env python3
"""A few passes of K2 + K3 on ONE partition of the windows (slice lists -> tally into the canonical half -> map ->
sweep of the kept lists) for rocprofv3: python3 scripts/k2k3_once.py [n_reads] [bins]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from lrbinner_amd import device as lrb
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
bins = int(sys.argv[2]) if len(sys.argv) > 2 else 32
L = 10_000
ctx = lrb.Context(0, use_torch_stream=True)
dev = torch.device("cuda")
codes, mask, co, mo, lens, words = bench.synth_packed(torch, n, L, 5, dev)
pr = lrb.PackedReads(codes, mask, co, mo, lens, n)
half = torch.zeros(lrb.K15_HALF_ENTRIES, dtype=torch.int32, device=dev)
wl = ctx.lists_alloc(pr, bins=bins)
h = torch.empty((n, bins), dtype=torch.int32, device=dev); s = torch.empty(n, dtype=torch.int32, device=dev)
m = None
for _ in range(3):
    half.zero_()
    ctx.lists_part_dev(pr, bins=bins, out=wl)
    ctx.lists_tally_dev(wl, half)
    m = ctx.cov_map_build_half_dev(half, 10, bins, map_t=m)
    ctx.cov_lists_sweep_dev(wl, m, bins, hist=h, sums=s)
torch.cuda.synchronize()
assert int(s.min().item()) == L - 14 and int(half.to(torch.int64).sum().item()) == n * (L - 14)
