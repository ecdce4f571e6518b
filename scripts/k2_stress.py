#!/usr/bin/env python3
"""Round 6: does the table stage ever lose a window?  One process of several that share ONE GPU (started side by side by
scripts/sessions/r06_k2stress.sh): m reads of 10 kb in resident batches, then again and again K2 through the product's
objects (HipCompute.k15_tally_half_many) -- with K1 + K8 text per batch in front of it on odd passes, as bench.py's text
pass and the sharded driver's phase A have it -- and the sum of the canonical half against m x (L - 14).
python3 scripts/k2_stress.py [passes=40] [m=20000]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from lrbinner_amd import dist as ld
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
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
    packed.append(ld._HipPacked(comp.ctx.packed_create_dev(seqs.data_ptr(), np.arange(nb + 1, dtype=np.uint64) * np.uint64(L), with_planes=2)))
    del seqs
torch.cuda.synchronize()
want = m * (L - 14)
bad = 0
ref = None
for p in range(passes):
    text = p % 2 == 1 and os.environ.get("STRESS_TEXT", "1") != "0"
    if text:
        for b in packed:
            b.kmer_text(4)
    half = comp.new_half()
    comp.k15_tally_half_many(packed, half, keep_bins=32)
    torch.cuda.synchronize()
    got = int(half.to(torch.int64).bitwise_and(0xFFFFFFFF).sum().item())
    if ref is None and got == want:
        ref = half.clone()
    if got != want:
        bad += 1
        where = ""
        if ref is not None:
            d = torch.nonzero(half != ref).flatten()
            where = f" {d.numel()} slots differ, first {d[:6].tolist()} got {half[d[:6]].tolist()} want {ref[d[:6]].tolist()}"
        print(f"[{tag}] pass {p} ({'text' if text else 'plain'}): half sums to {got}, {want - got} short{where}", flush=True)
    del half
print(f"[{tag}] {passes} passes, {bad} bad; partitions repeated by the count / part check: {comp.ctx.partition_retries()}", flush=True)
