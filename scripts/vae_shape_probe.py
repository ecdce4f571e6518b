#!/usr/bin/env python3
"""Step time of the fused VAE step (K7) for the network shapes of the BASELINE configs:
python scripts/vae_shape_probe.py  -> us per step by batch size for
  C1/C2  (10 + 32  -> 128-128 -> 4)     C3/C4/C5 (32 + 136 -> 128-128 -> 8)     k = 5 (32 + 512 -> 128-128 -> 8)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lrbinner_amd import ae_utils, device as lrb
from lrbinner_amd.vae_native import NativeTrainer

N = 300_000
for cov, prof, latent in ((10, 32, 4), (32, 136, 8), (32, 512, 8)):
    data = torch.rand(N, cov + prof, device="cuda")
    perm = torch.randperm(N, device="cuda")
    vae = ae_utils.VAE(cov, prof, latent_dims=latent, hidden_layers=[128, 128], device="cuda")
    w = ae_utils.h_params[str(prof)]
    ctx = lrb.Context(0, use_torch_stream=True)
    tr = NativeTrainer(ctx, vae, 8192, [w["e_cov_weight"], w["e_comp_weight"], w["kld_weight"]])
    tr.push()
    out = []
    for bs in (1024, 2048, 4096, 8192):
        nb = N // bs
        tr.train(data, perm, bs, nb); torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(3):
            tr.train(data, perm, bs, nb)
        torch.cuda.synchronize()
        out.append((time.time() - t0) / (3 * nb) * 1e6)
    print(f"{cov}+{prof} -> 128-128 -> {latent}: us/step at batch 1024/2048/4096/8192: " + " ".join(f"{v:.1f}" for v in out), flush=True)
    tr.close(); ctx.close()
