#!/usr/bin/env python3
"""K2 on several 400 k-read groups: part(g+1) on one stream WHILE split + tally(g) run on another (the part kernel is
VALU-bound, split / tally LDS- and HBM-bound), against the same launches in one stream.  Likewise K3's sweeps of two
groups side by side.  python3 scripts/k2_overlap_probe.py [groups]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from lrbinner_amd import device as lrb
import bench

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n, L, bins = 400_000, 10_000, 32
dev = torch.device("cuda")
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(sA):
    ctxA = lrb.Context(0, use_torch_stream=True)
with torch.cuda.stream(sB):
    ctxB = lrb.Context(0, use_torch_stream=True)
codes, mask, co, mo, lens, words = bench.synth_packed(torch, n * G, L, 5, dev)
groups = []
for g in range(G):
    a, b = g * n, (g + 1) * n
    pr = lrb.PackedReads(codes, mask, co[a:b + 1].contiguous(), mo[a:b + 1].contiguous(), lens[a:b].contiguous(), n)
    groups.append((pr, ctxA.lists_alloc(pr, bins=bins)))
half = torch.zeros(lrb.K15_HALF_ENTRIES, dtype=torch.int32, device=dev)
hist = torch.empty((n * G, bins), dtype=torch.int32, device=dev); sums = torch.empty(n * G, dtype=torch.int32, device=dev)
torch.cuda.synchronize()


def k2(overlap):
    half.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs = []
    for pr, wl in groups:
        with torch.cuda.stream(sA):
            ctxA.lists_part_dev(pr, bins=bins, out=wl)
            ev = torch.cuda.Event(); ev.record(sA); evs.append(ev)
        if overlap:
            with torch.cuda.stream(sB):
                sB.wait_event(ev)
                ctxB.lists_tally_dev(wl, half)
        else:
            with torch.cuda.stream(sA):
                ctxA.lists_tally_dev(wl, half)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


def k3(overlap, cmap):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i, (pr, wl) in enumerate(groups):
        st, cx = (sB, ctxB) if (overlap and i & 1) else (sA, ctxA)
        with torch.cuda.stream(st):
            cx.cov_lists_sweep_dev(wl, cmap, bins, hist=hist[i * n:(i + 1) * n], sums=sums[i * n:(i + 1) * n])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


for rep in range(3):
    a = k2(False)
    ref = half.clone()
    b = k2(True)
    same = bool(torch.equal(ref, half))
    cmap = ctxA.cov_map_build_half_dev(half, 10, bins)
    torch.cuda.synchronize()
    c = k3(False, cmap)
    href = hist.clone()
    d = k3(True, cmap)
    print(f"{G} groups of {n} reads: K2 one stream {a:.1f} ms, part(g+1) beside tally(g) {b:.1f} ms (same half: {same}); "
          f"K3 sweeps one stream {c:.1f} ms, two streams {d:.1f} ms (same histograms: {bool(torch.equal(href, hist))})", flush=True)
assert int(sums.min().item()) == L - 14
