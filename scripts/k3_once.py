"""One K3 pass over 100 k synthetic 10 kb reads against the 4 GiB table (cov_hist_kernel) and one against
the compact map (cov_hist_map_kernel), for rocprofv3 --pmc passes (scripts/prof_k3.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lrbinner_amd import device as lrb
from bench import synth_packed
dev = torch.device("cuda", 0)
ctx = lrb.Context(0, use_torch_stream=True)
n, L = 100_000, 10_000
codes, mask, co, mo, lens, words = synth_packed(torch, n, L, 1, dev)
pr = lrb.PackedReads(codes, mask, co, mo, lens, n)
table = torch.zeros(lrb.K15_ENTRIES, dtype=torch.int32, device=dev)
ctx.k15_accumulate_part_dev(pr, table, n * L)
ctx.k15_mirror_dev(table)
m = ctx.cov_map_build_dev(table, 10, 32)
hist = torch.empty((n, 32), dtype=torch.int32, device=dev); sums = torch.empty(n, dtype=torch.int32, device=dev)
for _ in range(2):
    ctx.cov_hist_dev(pr, table, 10, 32, hist=hist, sums=sums)
    ctx.cov_hist_map_dev(pr, m, 32, hist=hist, sums=sums)
torch.cuda.synchronize()
print("ok", int(sums[0]))
