#!/usr/bin/env python3
"""Round 6, second stress: WHICH step of the table stage loses a window when several processes share one GPU?  Per pass:
  lists   the window lists of the batches made ONCE in the workspaces, tallied TWICE into two halves: equal and wrong ->
          the lists are (part / order kernels); different -> the tally is;
  atomic  the same batches by one atomic a window (k15_accum_half_kernel: no lists) as the control.
Each against m x (L - 14) and against the first good half.  python3 scripts/k2_stress2.py [passes=30] [m=20000]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from lrbinner_amd import dist as ld, device as lrb
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 30
m = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
tag = os.environ.get("STRESS_TAG", "0")
L = 10_000
dev = torch.device("cuda")
comp = ld.HipCompute(0)
per = max(1, ld.PARSE_CHUNK_BYTES // (L + 8))
letters = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
g = torch.Generator(device=dev).manual_seed(777 + int(tag))
packed = []
for a in range(0, m, per):
    nb = min(per, m - a)
    seqs = letters[torch.randint(0, 4, (nb * L,), device=dev, generator=g, dtype=torch.int64)]
    packed.append(comp.ctx.packed_create_dev(seqs.data_ptr(), np.arange(nb + 1, dtype=np.uint64) * np.uint64(L), with_planes=2))
    del seqs
torch.cuda.synchronize()
want = m * (L - 14)
total = lambda h: int(h.to(torch.int64).bitwise_and(0xFFFFFFFF).sum().item())
ref = None
bad = {"lists_same_wrong": 0, "tally_differs": 0, "atomic_wrong": 0}
for p in range(passes):
    h1, h2, h3 = comp.new_half(), comp.new_half(), comp.new_half()
    wl = lrb.PackedLists(comp.ctx, packed, 32, workspace=True)
    wl.tally(h1.data_ptr())
    wl.tally(h2.data_ptr())
    wl.free()
    for rb in packed:
        rb.k15_accumulate_half(h3.data_ptr())
    torch.cuda.synchronize()
    t1, t2, t3 = total(h1), total(h2), total(h3)
    if ref is None and t1 == t2 == t3 == want and torch.equal(h1, h3):
        ref = h1.clone()
    if t3 != want or (ref is not None and not torch.equal(h3, ref)):
        bad["atomic_wrong"] += 1
        print(f"[{tag}] pass {p}: ATOMIC control sums to {t3} ({want - t3} short)", flush=True)
    if not torch.equal(h1, h2):
        bad["tally_differs"] += 1
        print(f"[{tag}] pass {p}: two tallies of ONE set of lists differ: {t1} / {t2} (want {want})", flush=True)
    elif t1 != want or (ref is not None and not torch.equal(h1, ref)):
        bad["lists_same_wrong"] += 1
        d = torch.nonzero(h1 != ref).flatten() if ref is not None else torch.zeros(0)
        print(f"[{tag}] pass {p}: both tallies equal and WRONG ({want - t1} short, {d.numel()} slots): the lists", flush=True)
    del h1, h2, h3
print(f"[{tag}] {passes} passes: {bad}", flush=True)
