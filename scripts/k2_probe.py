import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lrbinner_amd import device as lrb
from bench import synth_packed
dev = torch.device("cuda", 0)
ctx = lrb.Context(0, use_torch_stream=True)
L = 10_000
for n in ([int(a) for a in sys.argv[1:]] or [20_000, 100_000, 400_000]):
    codes, mask, co, mo, lens, words = synth_packed(torch, n, L, 1, dev)
    pr = lrb.PackedReads(codes, mask, co, mo, lens, n)
    table = torch.zeros(lrb.K15_ENTRIES, dtype=torch.int32, device=dev)
    def timed(fn, reps=3):
        fn(); torch.cuda.synchronize(); ts = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        return min(ts)
    td = timed(lambda: ctx.k15_accumulate_dev(pr, table))
    tp = timed(lambda: ctx.k15_accumulate_part_dev(pr, table, n * L))
    print(f"n={n}: direct {td:.2f} ms ({n/td*1e3/1e6:.2f} M reads/s)  partitioned {tp:.2f} ms ({n/tp*1e3/1e6:.2f} M reads/s)  x{td/tp:.2f}", flush=True)
    del codes, mask, pr, table
