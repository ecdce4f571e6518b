#!/usr/bin/env python3
"""The persistent XCD-local VAE step (LRB_VAE_PX=1): per-phase times of one step from the kernel's own stamps, the
control block (participants, time-out word), and the step time against the twelve-launch form.
python scripts/vae_px_probe.py [cov prof latent]"""
import os, sys, time
os.environ["LRB_VAE_PX"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lrbinner_amd import ae_utils, device as lrb
from lrbinner_amd.vae_native import NativeTrainer
cov, prof, lat = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (10, 32, 4)
vae = ae_utils.VAE(cov, prof, latent_dims=lat, hidden_layers=[128, 128], device="cuda")
w = ae_utils.h_params[str(prof)]
ctx = lrb.Context(0, use_torch_stream=True)
tr = NativeTrainer(ctx, vae, 8192, [w["e_cov_weight"], w["e_comp_weight"], w["kld_weight"]])
tr.push()
N = 200_000
data = torch.rand(N, cov + prof, device="cuda"); perm = torch.randperm(N, device="cuda")
B = 1024
tr.train(data, perm, B, 8); torch.cuda.synchronize()
tr.train(data, perm, B, 1); torch.cuda.synchronize()
ctl = tr.debug(90, 16 + 64 + 2 + 128).view(np.uint32)
stamps = ctl[82:].view(np.uint64)
print("arrivals per XCC", ctl[:8].tolist(), "time-out word", int(ctl[80]), "participants", int(ctl[81]))
n = int(np.count_nonzero(stamps))
d = np.diff(stamps[:n].astype(np.int64)) / 100.0
print("phase times of one step (us):", " ".join(f"{x:.1f}" for x in d), "| sum", f"{d.sum():.1f}")
for steps in (64, 195):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.train(data, perm, B, steps); torch.cuda.synchronize()
    print(f"{steps} steps in one launch: {(time.perf_counter() - t0) / steps * 1e6:.1f} us per step")
