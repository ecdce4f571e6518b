import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lrbinner_amd import ae_utils
rng = np.random.default_rng(0)
N = 432_333
centers = rng.random((8, 42))
lab = rng.integers(0, 8, N)
prof = centers[lab] + rng.normal(size=(N, 42)) * 0.05
cov, comp = prof[:, :10], prof[:, 10:]
for use_graph in (False, True):
    torch.manual_seed(0)
    vae = ae_utils.VAE(10, 32, latent_dims=4, hidden_layers=[128, 128], device="cuda")
    data = ae_utils.make_data(cov, comp, "cuda")
    def eval_loss():
        vae.eval()
        with torch.no_grad():
            mu, ls = vae._encode(data[:50000]); recon = vae._decode(mu)
            return float(vae.calc_loss(data[:50000], recon, mu, ls)[0])
    l0 = eval_loss()
    torch.cuda.synchronize(); t0 = time.time()
    vae.trainmodel(data, nepochs=4, batchsteps=[2], use_graph=use_graph)
    torch.cuda.synchronize(); dt = time.time() - t0
    steps = 2 * (N // 1024) + 2 * (N // 2048)
    print(f"graph={use_graph} loss {l0:.4f} -> {eval_loss():.4f}  {dt:.2f}s  {dt/steps*1e3:.3f} ms/step", flush=True)
