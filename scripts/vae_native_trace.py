"""Short run of the fused VAE step for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lrbinner_amd import ae_utils, device as lrb
from lrbinner_amd.vae_native import NativeTrainer
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cov, prof, lat = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (10, 32, 4)   # e.g. 32 136 8 for k = 4
vae = ae_utils.VAE(cov, prof, latent_dims=lat, hidden_layers=[128, 128], device="cuda")
w = ae_utils.h_params[str(prof)]
ctx = lrb.Context(0, use_torch_stream=True)
tr = NativeTrainer(ctx, vae, 8192, [w["e_cov_weight"], w["e_comp_weight"], w["kld_weight"]])
tr.push()
N = 200_000
data = torch.rand(N, cov + prof, device="cuda"); perm = torch.randperm(N, device="cuda")
tr.train(data, perm, bs, N // bs, use_graph=True); torch.cuda.synchronize()
