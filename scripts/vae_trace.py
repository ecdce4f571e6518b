"""One short graph-captured VAE training run for rocprofv3 --kernel-trace --stats:
which kernels make up a training step.  rocprofv3 ... -- python3 scripts/vae_trace.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lrbinner_amd import ae_utils
rng = np.random.default_rng(0)
N = 100_000
prof = rng.random((N, 42))
vae = ae_utils.VAE(10, 32, latent_dims=4, hidden_layers=[128, 128], device="cuda")
data = ae_utils.make_data(prof[:, :10], prof[:, 10:], "cuda")
vae.trainmodel(data, nepochs=int(sys.argv[1]) if len(sys.argv) > 1 else 3, batchsteps=[], use_graph=True)
torch.cuda.synchronize()
