#!/usr/bin/env python3
"""Round 6: the same question for the other LDS-tallying kernels -- K1 at k = 3 / 4 / 5 (every row must sum to L - k + 1 and
equal the first pass's) and K3 as a sweep of resident lists and by the default route (every read's histogram must sum to
L - 14 and equal the first pass's) -- several processes sharing one GPU.  python3 scripts/k13_stress.py [passes=40] [m=20000]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from lrbinner_amd import device as lrb
from bench import synth_packed
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
m = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
tag = os.environ.get("STRESS_TAG", "0")
L = 10_000
dev = torch.device("cuda")
ctx = lrb.Context(0, use_torch_stream=True)
codes, mask, co, mo, lens, words = synth_packed(torch, m, L, 999 + int(tag), dev)
pr = lrb.PackedReads(codes, mask, co, mo, lens, m)
ctx.make_planes(pr); ctx.make_planes_t(pr, sort=True); ctx.make_codes_t(pr, sort=True)
half = torch.zeros(lrb.K15_HALF_ENTRIES, dtype=torch.int32, device=dev)
ctx.k15_accumulate_half_dev(pr, half)
cmap = ctx.cov_map_build_half_dev(half, 10, 32)
wl = ctx.lists_part_dev(pr, bins=32)
ref = {}
bad = {"k1_k3": 0, "k1_k4": 0, "k1_k5": 0, "k3_lists": 0, "k3_default": 0}
for p in range(passes):
    outs = {"k1_k3": ctx.kmer_counts3t_dev(pr), "k1_k4": ctx.kmer_counts4t_dev(pr, k=4), "k1_k5": ctx.kmer_counts4t_dev(pr, k=5)}
    h1 = torch.empty((m, 32), dtype=torch.int32, device=dev); s1 = torch.empty(m, dtype=torch.int32, device=dev)
    ctx.cov_lists_sweep_dev(wl, cmap, 32, hist=h1, sums=s1)
    h2 = torch.empty((m, 32), dtype=torch.int32, device=dev); s2 = torch.empty(m, dtype=torch.int32, device=dev)
    ctx.cov_hist_sweep_dev(pr, cmap, 32, hist=h2, sums=s2)
    torch.cuda.synchronize()
    for k_, kk in (("k1_k3", 3), ("k1_k4", 4), ("k1_k5", 5)):
        rows = outs[k_].sum(dim=1)
        ok = bool((rows == L - kk + 1).all()) and (k_ not in ref or torch.equal(outs[k_], ref[k_]))
        if not ok:
            bad[k_] += 1
            print(f"[{tag}] pass {p}: {k_}: {int((rows != L - kk + 1).sum())} rows with a wrong sum", flush=True)
        ref.setdefault(k_, outs[k_].clone())
    for k_, h, s_ in (("k3_lists", h1, s1), ("k3_default", h2, s2)):
        ok = bool((s_ == L - 14).all()) and bool((h.sum(dim=1) == L - 14).all()) and (k_ not in ref or torch.equal(h, ref[k_]))
        if not ok:
            bad[k_] += 1
            print(f"[{tag}] pass {p}: {k_}: {int((h.sum(dim=1) != L - 14).sum())} reads with a wrong histogram sum, equal to the first pass: {k_ in ref and torch.equal(h, ref[k_])}", flush=True)
        ref.setdefault(k_, h.clone())
print(f"[{tag}] {passes} passes: {bad}; partitions repeated: {ctx.partition_retries()}", flush=True)
