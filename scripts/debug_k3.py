import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lrbinner_amd import device
from oracle import oracle as orc
ctx = device.Context(0, use_torch_stream=True)
reads = [b"A" * 40, b"ACG" * 20, b"C" * 33, b"AAC" + b"A" * 61, b"ACGTTGCATGCATGACTGAC" * 7,
         bytes(np.random.default_rng(1).choice(np.frombuffer(b"ACGT", np.uint8), 3000))]
buf, offs = orc.concat(reads)
pr = ctx.pack(torch.from_numpy(buf).cuda(), offs)
ctx.make_planes(pr)
exp, _ = orc.count_kmers(buf, offs, 3)
got = ctx.kmer_counts3_dev(pr, mode=2).cpu().numpy().view(np.uint32)
for i in range(len(reads)):
    print(i, "ok" if np.array_equal(got[i], exp[i]) else "BAD")
    if not np.array_equal(got[i], exp[i]):
        print(" exp", exp[i].tolist()); print(" got", got[i].tolist())
