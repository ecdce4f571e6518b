#!/usr/bin/env python3
"""Where a C4-shaped rank's table and coverage phases spend their time beyond their kernels: the phases of bench.py's
c4_phases (default route) on m reads, wall time per phase; run under `rocprofv3 --kernel-trace --stats` the kernels' sum
stands beside it.  python3 scripts/c4_gap_probe.py [m_reads]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from lrbinner_amd import dist as ld
m = int(sys.argv[1]) if len(sys.argv) > 1 else 2_500_000
L = 10_000
dev = torch.device("cuda")
comp = ld.HipCompute(0)
per = max(1, ld.PARSE_CHUNK_BYTES // (L + 8))
letters = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
g = torch.Generator(device=dev).manual_seed(777)
packed = []
for a in range(0, m, per):
    nb = min(per, m - a)
    seqs = letters[torch.randint(0, 4, (nb * L,), device=dev, generator=g, dtype=torch.int64)]
    packed.append(ld._HipPacked(comp.ctx.packed_create_dev(seqs.data_ptr(), np.arange(nb + 1, dtype=np.uint64) * np.uint64(L), with_planes=2)))
    del seqs
items = list(enumerate(packed))
torch.cuda.synchronize()
for rep in range(3):
    half = comp.new_half()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    kept = comp.k15_tally_half_many(packed, half, keep_bins=32)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    table = comp.table_from_half(half)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    for _ in comp.cov_hist_groups(items, table, 10, 32, kept=kept):
        pass
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"pass {rep}: K2 {1e3 * (t1 - t0):.1f} ms, expand {1e3 * (t2 - t1):.1f} ms, K3 {1e3 * (t3 - t2):.1f} ms", flush=True)
    del table, half
