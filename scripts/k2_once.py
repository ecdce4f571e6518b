"""One partitioned K2 accumulate of 400 k synthetic 10 kb reads (4.0e9 windows, one group), twice,
for rocprofv3 passes (scripts/prof_k2.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lrbinner_amd import device as lrb
from bench import synth_packed
dev = torch.device("cuda", 0)
ctx = lrb.Context(0, use_torch_stream=True)
n, L = 400_000, 10_000
codes, mask, co, mo, lens, words = synth_packed(torch, n, L, 1, dev)
pr = lrb.PackedReads(codes, mask, co, mo, lens, n)
table = torch.zeros(lrb.K15_ENTRIES, dtype=torch.int32, device=dev)
for _ in range(2):
    ctx.k15_accumulate_part_dev(pr, table, n * L)
torch.cuda.synchronize()
print("ok", int(table[:1000].sum()))
