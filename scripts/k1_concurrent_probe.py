"""Probe: do the LDS-histogram kernel and the bit-plane kernel add up when they run at the
same time on two HIP streams (disjoint halves of the reads)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lrbinner_amd import device as lrb
from bench import synth_packed
dev = torch.device("cuda", 0)
n, L = 1_000_000, 10_000
codes, mask, co, mo, lens, words = synth_packed(torch, n, L, 1, dev)
base = lrb.Context(0, use_torch_stream=True)
pr = lrb.PackedReads(codes, mask, co, mo, lens, n)
base.make_planes(pr); torch.cuda.synchronize()
A, B = lrb.Context(0), lrb.Context(0)       # two contexts = two HIP streams
out = torch.empty((n, 32), dtype=torch.int32, device=dev)
def sub(lo, hi):
    return lrb.PackedReads(pr.codes, pr.mask, pr.code_off[lo:hi + 1].contiguous(), pr.mask_off[lo:hi + 1].contiguous(),
                           pr.lens[lo:hi].contiguous(), hi - lo, pr.planes)
def run(frac, reps=10):
    n1 = int(n * frac)
    p1, p2 = sub(0, n1), sub(n1, n)
    o1, o2 = out[:n1], out[n1:]
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps + 2):
        t0 = time.perf_counter()
        if n1: A.kmer_counts3_dev(p1, mode=1, out=o1)
        if n - n1: B.kmer_counts3_dev(p2, mode=2, out=o2)
        A.sync(); B.sync()
        ts.append(time.perf_counter() - t0)
    return np.median(ts[2:]) * 1e3
for f in (0.0, 1.0, 0.3, 0.4, 0.5, 0.6):
    print(f"LDS fraction {f:.1f}: {run(f):.3f} ms per 1M reads", flush=True)
assert int(out[:1000].sum(dim=1).min()) == L - 2 and int(out[-1000:].sum(dim=1).min()) == L - 2
