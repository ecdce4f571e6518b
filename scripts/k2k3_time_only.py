#!/usr/bin/env python3
"""As k2k3_once.py without the result checks (for timing experiments that break the lists on purpose)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from lrbinner_amd import device as lrb
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
ctx = lrb.Context(0, use_torch_stream=True)
dev = torch.device("cuda")
codes, mask, co, mo, lens, words = bench.synth_packed(torch, n, 10_000, 5, dev)
pr = lrb.PackedReads(codes, mask, co, mo, lens, n)
wl = ctx.lists_alloc(pr, bins=32)
for _ in range(3):
    ctx.lists_part_dev(pr, bins=32, out=wl)
torch.cuda.synchronize()
