#!/usr/bin/env python3
"""K4 (seed_hist_kernel): exactness against torch.histc of the library's own single-seed distances over row lengths and seed
counts (incl. seeds repeated, fewer seeds than a workgroup holds, run-time row lengths), then kernel time at C1's size.
python3 scripts/k4_seed_hist_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from lrbinner_amd import device as lrb

ctx = lrb.Context(0, use_torch_stream=True)
g = torch.Generator(device="cuda").manual_seed(4)
SHAPES = ((4, 432_333, 1000), (8, 200_000, 300), (2, 50_000, 64), (3, 10_007, 257), (10, 30_000, 130), (32, 20_000, 70), (64, 9_000, 5), (4, 100, 7))
if os.environ.get("K4_ONLY_C1"):   # for rocprofv3: one shape, so that the kernel averages mean something
    SHAPES = SHAPES[:1]
for d, n, S in SHAPES:
    if os.environ.get("K4_BLOBS"):   # eight clusters, as a metagenome's latents: an eighth of the pairs within 0.3
        cen = torch.randn((8, d), device="cuda", generator=g) * 2
        M = cen[torch.randint(0, 8, (n,), device="cuda", generator=g)] + torch.randn((n, d), device="cuda", generator=g) * 0.15
    else:
        M = torch.randn((n, d), device="cuda", generator=g) * 0.2 + torch.linspace(-1, 1, d, device="cuda")
    M = (M / M.norm(dim=1, keepdim=True) * (0.5 ** 0.5)).contiguous()
    seeds = torch.randint(0, n, (S,), device="cuda", generator=g, dtype=torch.int64)
    seeds[-1] = seeds[0]            # a repeated seed
    h = ctx.seed_hist_dev(M, seeds).cpu().numpy().view(np.uint32)
    bad = 0
    for j in list(range(min(S, 12))) + [S - 1]:
        dd = ctx.seed_dist_dev(M, int(seeds[j]))
        ref = torch.histc(dd.cpu(), 60, 0, 0.3).numpy()
        bad += int(not np.array_equal(h[j].astype(np.float32), ref))
    t = []
    for _ in range(5):
        torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            ctx.seed_hist_dev(M, seeds)
        b.record(); torch.cuda.synchronize(); t.append(a.elapsed_time(b) / 10)
    inr = float(h[: min(S, 64)].sum()) / (min(S, 64) * n)
    print(f"dims {d:2d} n {n:7d} seeds {S:4d} ({inr:.2f} of the pairs within 0.3): mismatching histograms {bad}; {min(t):.3f} ms per call (incl. the memset), "
          f"{n * S / (min(t) * 1e-3) / 1e12:.2f} T pairs/s", flush=True)
