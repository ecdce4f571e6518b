#!/usr/bin/env python3
"""Round 6, third stress: the partition of one set of resident reads TWICE in a row (lrb_k15_lists_part_dev), several
processes sharing one GPU; after each partition the level-1 lists (the part kernel's output, workspace slot 9) and the
final lists (the order kernel's output) are reduced to one 64-bit hash sum per (group, slice) / per (group, bucket) --
order inside a run is the atomics' and may differ, the multiset may not.  Which kernel's output differs between two
partitions of the same reads?  python3 scripts/k2_stress3.py [passes=30] [m=20000]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from lrbinner_amd import device as lrb
from lrbinner_amd._lib import call, vp
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 30
m = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
tag = os.environ.get("STRESS_TAG", "0")
L = 10_000
dev = torch.device("cuda")
ctx = lrb.Context(0, use_torch_stream=True)
sys.path.insert(0, ROOT)
from bench import synth_packed
codes, mask, co, mo, lens, words = synth_packed(torch, m, L, 4242 + int(tag), dev)
pr = lrb.PackedReads(codes, mask, co, mo, lens, m)
SLICES, SUBS, BUCKETS = 256, 64, 16384
K = torch.tensor(-7046029254386353131, dtype=torch.int64, device=dev)   # 0x9E3779B97F4A7C15


def seg_sums(values_i32, starts, ends, nseg):
    """sum of hash(entry) per segment [starts[i], ends[i]) of a flat int32 tensor (segments tile it in order)."""
    counts = (ends - starts).clamp(min=0)
    ids = torch.repeat_interleave(torch.arange(nseg, device=dev), counts)
    # (a group's region is sized by its mask words, its entries end before the next group's begin: positions explicitly)
    first = torch.cumsum(counts, 0) - counts
    pos = torch.repeat_interleave(starts - first, counts) + torch.arange(int(counts.sum()), device=dev)
    h = values_i32[pos].to(torch.int64) * K
    out = torch.zeros(nseg, dtype=torch.int64, device=dev)
    out.index_add_(0, ids, h)
    return out


def snapshot(wl):
    torch.cuda.synchronize()
    ng = wl.ngroups
    bounds = wl.bounds[: ng * (BUCKETS + 1)].view(ng, BUCKETS + 1).to(torch.int64)
    gbase = wl.gbase.to(torch.int64)
    total = int(gbase[ng] - gbase[0])
    p, b = vp(), C.c_uint64(0)
    call("lrb_ctx_ws_info", ctx._h, 9, C.byref(p), C.byref(b))
    assert b.value >= 4 * total, (b.value, total)          # one chunk: the whole level-1 set is there
    l1 = torch.empty(total, dtype=torch.int32, device=dev)
    C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(l1.data_ptr()), C.c_void_p(p.value), C.c_size_t(4 * total), C.c_int(3))
    fin = wl.lists[:total]
    # level-1: per (group, slice) the run [bounds[g][64 s], bounds[g][64 (s + 1)]) from the group's base
    s0 = bounds[:, 0:BUCKETS:SUBS]                                                   # [ng][256]
    s1 = torch.cat([bounds[:, SUBS:BUCKETS:SUBS], bounds[:, BUCKETS:BUCKETS + 1]], 1)  # [ng][256]
    base = (gbase[:ng] - gbase[0]).unsqueeze(1)
    l1s = seg_sums(l1, (base + s0).flatten(), (base + s1).flatten(), ng * SLICES)
    fs = seg_sums(fin, (base + bounds[:, :BUCKETS]).flatten(), (base + bounds[:, 1:]).flatten(), ng * BUCKETS)
    return bounds.clone(), l1s, fs


wl = ctx.lists_alloc(pr, bins=32)
ctx.lists_part_dev(pr, bins=32, out=wl)
ref = snapshot(wl)
bad = {"bounds": 0, "level1": 0, "final_only": 0}
for p_ in range(passes):
    ctx.lists_part_dev(pr, bins=32, out=wl)
    got = snapshot(wl)
    b_bad = not torch.equal(got[0], ref[0])
    l_bad = not torch.equal(got[1], ref[1])
    f_bad = not torch.equal(got[2], ref[2])
    if b_bad:
        bad["bounds"] += 1
        w = torch.nonzero(got[0] != ref[0])
        words = [(int(g_), int(x_)) for g_, x_ in w[:8].tolist()]
        slice_words = int(((w[:, 1] % SUBS) == 0).sum())
        print(f"[{tag}] pass {p_}: BOUNDS differ in {w.shape[0]} words ({slice_words} of them slice starts / group totals): (group, word) {words}; "
              f"got {[int(got[0][g_, x_]) for g_, x_ in words]} want {[int(ref[0][g_, x_]) for g_, x_ in words]}", flush=True)
    if l_bad:
        bad["level1"] += 1
        d = torch.nonzero(got[1] != ref[1]).flatten()
        print(f"[{tag}] pass {p_}: LEVEL-1 lists differ (part kernel) in {d.numel()} (group, slice) runs: "
              f"{[(int(x) // SLICES, int(x) % SLICES) for x in d[:6].tolist()]}", flush=True)
    if f_bad:
        bad["final_only"] += (not l_bad)
        d = torch.nonzero(got[2] != ref[2]).flatten()
        print(f"[{tag}] pass {p_}: FINAL lists differ in {d.numel()} (group, bucket) runs: "
              f"{[(int(x) // BUCKETS, (int(x) % BUCKETS) // SUBS, int(x) % SUBS) for x in d[:6].tolist()]} (level-1 {'differs' if l_bad else 'equal'})", flush=True)
print(f"[{tag}] {passes} passes, R {wl.R}, groups {wl.ngroups}: {bad}", flush=True)
