#!/usr/bin/env python3
"""What would ORDER inside the slice lists buy K3's sweep?  The sweep issues one L2 request per window (a byte gather
from the 2 MB map slice in the L2): 214 G requests/s, 79 % of what L2-hit gathers reach.  With R reads per group a
(group, slice) list holds R x 39 entries over the slice's 16,384 lines of 128 bytes -- 5 entries per line at R = 2,048.
If the entries of a list are ordered by offset, the lanes of a load instruction fall into few lines and the texture
unit sends one request per line.  This probe builds the lists as usual, then sorts every (group, slice) list on the
side (torch) -- fully by offset, or only by its top 6 bits (the 32 KB sub-slice: order the CU's L1 could exploit) --
and times the unchanged sweep kernel on them.  Histograms must not change.
python3 scripts/k3_order_probe.py [reads per group ...]     (env LRB_K3_SWEEP_READS is set per run)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

if len(sys.argv) > 1 and sys.argv[1] != "--one":
    for R in sys.argv[1:]:
        subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=dict(os.environ, LRB_K3_SWEEP_READS=R))
    sys.exit(0)

import torch
from lrbinner_amd import device as lrb
import bench

n, L, bins = 400_000, 10_000, 32
dev = torch.device("cuda")
ctx = lrb.Context(0, use_torch_stream=True)
codes, mask, co, mo, lens, words = bench.synth_packed(torch, n, L, 5, dev)
pr = lrb.PackedReads(codes, mask, co, mo, lens, n)
half = torch.zeros(lrb.K15_HALF_ENTRIES, dtype=torch.int32, device=dev)
wl = ctx.lists_alloc(pr, bins=bins)
R, G = wl.R, wl.ngroups
ctx.lists_part_dev(pr, bins=bins, out=wl)
ctx.lists_tally_dev(wl, half, n * L)
cmap = ctx.cov_map_build_half_dev(half, 10, bins)
del half
hist = torch.empty((n, bins), dtype=torch.int32, device=dev); sums = torch.empty(n, dtype=torch.int32, device=dev)


def sweep_ms(reps=3):
    ctx.cov_lists_sweep_dev(wl, cmap, bins, hist=hist, sums=sums)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.cov_lists_sweep_dev(wl, cmap, bins, hist=hist, sums=sums)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


base = sweep_ms()
ref = hist.clone()
first_word = int(mo[0].item())
sizes = wl.sizes.view(G, 256).to(torch.int64)
out = [f"{R} reads per group ({G} groups): lists as the part kernel leaves them {base:.1f} ms"]
orig = wl.lists.clone()
for name, keep_bits in (("ordered by 32 KB sub-slice (top 6 bits of the offset)", 6), ("ordered by offset", 21)):
    wl.lists.copy_(orig)
    for g in range(G):
        r0 = g * R
        w0 = int(mo[min(r0, n)].item())
        base_off = (w0 - first_word) * 32
        sz = sizes[g]
        tot = int(sz.sum().item())
        if tot == 0:
            continue
        seg = wl.lists[base_off:base_off + tot]
        slice_id = torch.repeat_interleave(torch.arange(256, device=dev), sz)
        off = (seg.to(torch.int64) & 0x1FFFFF) >> (21 - keep_bits)
        key = (slice_id << keep_bits) | off
        order = torch.sort(key, stable=True)[1]
        seg.copy_(seg[order])
    ms = sweep_ms()
    same = bool(torch.equal(ref, hist))
    out.append(f"  {name}: {ms:.1f} ms (same histograms: {same})")
print("\n".join(out), flush=True)
