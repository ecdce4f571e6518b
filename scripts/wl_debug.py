#!/usr/bin/env python3
"""Stage-by-stage check of the window-list route on random ACGT reads: python3 scripts/wl_debug.py [n_reads] [L] [R]
 1. lists + bounds against a numpy computation of every window's pair index (per group and bucket, as multisets)
 2. tally against lrb_k15_accumulate_half_dev
 3. sweep against the gather kernel"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from lrbinner_amd import device as lrb

n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
L = int(sys.argv[2]) if len(sys.argv) > 2 else 700
if len(sys.argv) > 3:
    os.environ["LRB_K3_SWEEP_READS"] = sys.argv[3]
rng = np.random.default_rng(3)
lens = rng.integers(max(15, L // 2), L + 1, size=n)
reads = [rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(l)).tobytes() for l in lens]
buf = np.frombuffer(b"".join(reads), np.uint8)
offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
ctx = lrb.Context(0, use_torch_stream=True)
pr = ctx.pack(torch.from_numpy(buf.copy()).cuda(), offs)


def pair_index(seq):
    """h of every 15-mer of an ACGT byte string (numpy)"""
    c = ((np.frombuffer(seq, np.uint8) >> 1) & 3).astype(np.int64)
    m = len(c) - 14
    val = np.zeros(m, np.int64)
    rc = np.zeros(m, np.int64)
    for i in range(15):
        val |= c[i:i + m] << (2 * (14 - i))
        rc |= (c[i:i + m] ^ 2) << (2 * i)
    x = np.where(val & 0x8000, rc, val)
    return ((x >> 16) << 15) | (x & 0x7FFF)


wl = ctx.lists_part_dev(pr, bins=32)
torch.cuda.synchronize()
R, G = wl.R, wl.ngroups
print(f"n {n} L {L} R {R} groups {G}")
bounds = wl.bounds[: G * 16385].view(G, 16385).cpu().numpy().astype(np.int64)
gbase = wl.gbase[: G + 1].cpu().numpy()
lists = wl.lists.cpu().numpy().view(np.uint32)
bad = 0
check = list(range(G)) if G <= 300 and n * L <= 5e8 else [0, 1, G // 2, G - 1]
for g in check:
    b = bounds[g]
    if b[0] != 0 or (np.diff(b) < 0).any():
        print(f"group {g}: bounds not monotone / not starting at 0: first {b[:5]} total {b[-1]}")
        bad += 1
        continue
    want = []
    for r in range(g * R, min((g + 1) * R, n)):
        h = pair_index(reads[r])
        want.append(((r - g * R) << 40) | h)
    want = np.sort(np.concatenate(want)) if want else np.zeros(0, np.int64)
    if b[-1] != len(want):
        print(f"group {g}: {b[-1]} entries, want {len(want)}")
        bad += 1
        continue
    e = lists[gbase[g]: gbase[g] + b[-1]].astype(np.int64)
    bucket = np.repeat(np.arange(16384), np.diff(b))
    if ((e >> 15) & 63 != (bucket & 63)).any():
        k = int(np.nonzero((e >> 15) & 63 != (bucket & 63))[0][0])
        print(f"group {g}: entry {k} = {e[k]:#x} sits in bucket {bucket[k]} (sub {bucket[k] & 63})")
        bad += 1
        continue
    got = np.sort(((e >> 21) << 40) | ((bucket >> 6) << 21) | (e & 0x1FFFFF))
    if not np.array_equal(got, want):
        d = int(np.nonzero(got != want)[0][0])
        print(f"group {g}: multiset differs at sorted position {d}: got {got[d]:#x} want {want[d]:#x}")
        gu, gc = np.unique(got, return_counts=True)
        wu, wc = np.unique(want, return_counts=True)
        allv = np.union1d(gu, wu)
        gcount = np.zeros(len(allv), np.int64); gcount[np.searchsorted(allv, gu)] = gc
        wcount = np.zeros(len(allv), np.int64); wcount[np.searchsorted(allv, wu)] = wc
        extra = allv[gcount > wcount]; missing = allv[gcount < wcount]
        print(f"   extra {len(extra)} missing {len(missing)}; extra reads {np.unique(extra >> 40)[:10]} missing reads {np.unique(missing >> 40)[:10]}")
        print("   extra values", [hex(int(x)) for x in extra[:12]], "missing", [hex(int(x)) for x in missing[:12]])
        print(f"   extra slices {np.unique((extra >> 21) & 0xFF)[:20]}  missing slices {np.unique((missing >> 21) & 0xFF)[:20]}")
        bad += 1
print("lists:", "OK" if not bad else f"{bad} bad groups")
half = torch.zeros(lrb.K15_HALF_ENTRIES, dtype=torch.int32, device="cuda")
ctx.lists_tally_dev(wl, half)
half2 = torch.zeros(lrb.K15_HALF_ENTRIES, dtype=torch.int32, device="cuda")
ctx.k15_accumulate_half_dev(pr, half2)
torch.cuda.synchronize()
ne = int((half != half2).sum().item())
print("tally:", "OK" if ne == 0 else f"{ne} entries differ; sums {int(half.sum())} {int(half2.sum())}")
cmap = ctx.cov_map_build_half_dev(half2, 2, 32)
h0, s0 = ctx.cov_hist_map_dev(pr, cmap, 32)
h1, s1 = ctx.cov_lists_sweep_dev(wl, cmap, 32)
torch.cuda.synchronize()
nr = int((h0 != h1).any(dim=1).sum().item())
print("sweep:", "OK" if nr == 0 and torch.equal(s0[:n], s1[:n]) else f"{nr} rows differ")
if nr:
    r = int(torch.nonzero((h0 != h1).any(dim=1))[0].item())
    print(" first bad row", r, "group", r // R, "\n  want", h0[r].tolist(), "\n  got ", h1[r].tolist())
