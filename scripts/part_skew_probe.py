#!/usr/bin/env python3
"""The window partition (count + part + order) on reads that are NOT uniform: AT-rich, half the reads one homopolymer,
one base in a hundred not ACGT, short reads.  One line per input, each in its own process: python3 scripts/part_skew_probe.py
[n_reads].  (profiles/r05_part_ring_ab.txt holds this probe's lines for the two forms of the part kernel while both existed:
commit 0065ff3, switch LRB_WL_PART_RING; the ring form is the only one since.)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000

def child(kind):
    import numpy as np, torch
    from lrbinner_amd import device as lrb
    ctx = lrb.Context(0, use_torch_stream=True)
    dev = torch.device("cuda")
    L = 150 if kind == "short" else 10_000
    nn = n * 20 if kind == "short" else n
    ncw = -(-L // 16); words = ((ncw + 3) // 4) * 4 + 4
    nmw = -(-L // 32); mwords = ((nmw + 3) // 4) * 4 + 4
    g = torch.Generator(device=dev).manual_seed(7)
    probs = {"uniform": [.25, .25, .25, .25], "at_rich": [.4, .1, .4, .1], "homopolymer": [.25, .25, .25, .25],
             "n_bases": [.25, .25, .25, .25], "short": [.25, .25, .25, .25]}[kind]
    cdf = torch.tensor(np.cumsum(probs)[:3], device=dev, dtype=torch.float32)
    codes = torch.zeros((nn, words), dtype=torch.int64, device=dev)
    mask = torch.zeros((nn, mwords), dtype=torch.int64, device=dev)
    step = max(1, (1 << 27) // (ncw * 16))
    for r0 in range(0, nn, step):
        r1 = min(nn, r0 + step)
        u = torch.rand((r1 - r0, ncw * 16), device=dev, generator=g)
        b = (u[..., None] >= cdf).sum(-1).to(torch.int64)          # A0 C1 T2 G3
        if kind == "homopolymer":
            b[::2] = 0
        b[:, L:] = 0
        sh = (30 - 2 * (torch.arange(ncw * 16, device=dev) % 16)).to(torch.int64)
        codes[r0:r1, :ncw] = (b << sh).view(r1 - r0, ncw, 16).sum(-1)
        ok = torch.ones((r1 - r0, nmw * 32), dtype=torch.int64, device=dev)
        ok[:, L:] = 0
        if kind == "n_bases":
            ok &= (torch.rand(ok.shape, device=dev, generator=g) >= 0.01).to(torch.int64)
        shm = (31 - (torch.arange(nmw * 32, device=dev) % 32)).to(torch.int64)
        mask[r0:r1, :nmw] = (ok << shm).view(r1 - r0, nmw, 32).sum(-1)
    codes = codes.to(torch.int32).view(-1); mask = mask.to(torch.int32).view(-1)   # (wraps to the same 32 bits)
    co = torch.arange(nn + 1, dtype=torch.int64, device=dev) * words
    mo = torch.arange(nn + 1, dtype=torch.int64, device=dev) * mwords
    lens = torch.full((nn,), L, dtype=torch.int32, device=dev)
    pr = lrb.PackedReads(codes, mask, co, mo, lens, nn)
    wl = ctx.lists_alloc(pr, bins=32)
    half = torch.zeros(lrb.K15_HALF_ENTRIES, dtype=torch.int32, device=dev)
    ctx.lists_part_dev(pr, bins=32, out=wl); ctx.lists_tally_dev(wl, half)
    total = int(half.to(torch.int64).sum().item())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ms = []
    for _ in range(4):
        e0.record(); ctx.lists_part_dev(pr, bins=32, out=wl); e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    print(f"{kind:12s}: {nn} reads x {L}, {total} windows tallied, count + part + order {min(ms):8.3f} ms (best of 4)", flush=True)

if len(sys.argv) > 2:
    child(sys.argv[2])
else:
    for kind in ("uniform", "at_rich", "homopolymer", "n_bases", "short"):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), str(n), kind], capture_output=True, text=True, timeout=600)
        print((r.stdout.strip().splitlines() or ["(no output) " + r.stderr[-300:]])[-1], flush=True)
