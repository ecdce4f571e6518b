#!/usr/bin/env python3
"""A few launches of the K3 sweep (and of the gather form) for rocprofv3: python3 scripts/k3_sweep_once.py [n_reads] [bins]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from lrbinner_amd import device as lrb
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
bins = int(sys.argv[2]) if len(sys.argv) > 2 else 32
ctx = lrb.Context(0, use_torch_stream=True)
dev = torch.device("cuda")
codes, mask, co, mo, lens, words = bench.synth_packed(torch, n, 10_000, 5, dev)
pr = lrb.PackedReads(codes, mask, co, mo, lens, n)
table = torch.zeros(lrb.K15_ENTRIES, dtype=torch.int32, device=dev)
ctx.k15_accumulate_part_dev(pr, table, n * 10_000)
ctx.k15_mirror_dev(table)
m = ctx.cov_map_build_dev(table, 10, bins)
h = torch.empty((n, bins), dtype=torch.int32, device=dev); s = torch.empty(n, dtype=torch.int32, device=dev)
for _ in range(3):
    ctx.cov_hist_sweep_dev(pr, m, bins, hist=h, sums=s)
ctx.cov_hist_map_dev(pr, m, bins, hist=h, sums=s)
torch.cuda.synchronize()
