import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from lrbinner_amd import ae_utils, device as lrb
from lrbinner_amd.vae_native import NativeTrainer
vae = ae_utils.VAE(10, 32, latent_dims=4, hidden_layers=[128, 128], device="cuda")
w = ae_utils.h_params["32"]
ctx = lrb.Context(0, use_torch_stream=True)
tr = NativeTrainer(ctx, vae, 8192, [w["e_cov_weight"], w["e_comp_weight"], w["kld_weight"]])
tr.push()
N = 200_000
data = torch.rand(N, 42, device="cuda"); perm = torch.randperm(N, device="cuda")
for bs in (1024, 2048):
    tr.train(data, perm, bs, 20, use_graph=True); torch.cuda.synchronize()
    tiles = (42 + 15) // 16 + 8 + 8 + 1 + 8 + 8   # out, dec1, dec0, heads, enc1, enc0
    nwg = min(tiles * (bs // 128), 1024)
    d = tr.debug(90, nwg * 8 * 2).view(np.uint64).reshape(nwg, 8).astype(np.int64)
    t0 = d[:, 0].min()
    rel = (d - t0) * 10
    print("batch", bs, "WGs", nwg, "tiles", tiles)
    print(" start spread ns: min/med/max", rel[:, 0].min(), int(np.median(rel[:, 0])), rel[:, 0].max())
    for i in range(1, 8):
        dd = (d[:, i] - d[:, i - 1]) * 10
        print(f"  phase {i-1}->{i}: min {dd.min()} med {int(np.median(dd))} max {dd.max()}")
    print("  kernel span ns", rel[:, 7].max())
    # per layer (tile index = wg % tiles)
    x = np.arange(nwg) % tiles
    for name, lo, hi in (("out", 0, 3), ("dec1", 3, 11), ("dec0", 11, 19), ("heads", 19, 20), ("enc1", 20, 28), ("enc0", 28, 36)):
        m = (x >= lo) & (x < hi)
        if m.any():
            print(f"   {name}: WG duration med {int(np.median(rel[m, 7] - rel[m, 0]))} end max {rel[m, 7].max()}  desc wait med {int(np.median((d[m,1]-d[m,0])*10))}")
