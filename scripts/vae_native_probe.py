#!/usr/bin/env python3
"""The fused VAE step (K7) against torch autograd on the same batch, masks and eps: loss terms,
every parameter gradient, the parameters and running statistics after Adam steps; then timing
against the graph-captured torch step.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as F
from lrbinner_amd import ae_utils, device as lrb
from lrbinner_amd.vae_native import NativeTrainer, keep_mask

torch.manual_seed(0)
rng = np.random.default_rng(0)
cov_size, prof_size, hidden, latent = 10, 32, [128, 128], 4
N, B, seed = 20000, 1024, 12345
X = rng.random((N, cov_size + prof_size)).astype(np.float32)
X[:, 3] = 0.0
data = torch.from_numpy(X).cuda()
vae = ae_utils.VAE(cov_size, prof_size, latent_dims=latent, hidden_layers=hidden, device="cuda")
w = ae_utils.h_params[str(prof_size)]
weights = [w["e_cov_weight"], w["e_comp_weight"], w["kld_weight"]]
ctx = lrb.Context(0, use_torch_stream=True)
tr = NativeTrainer(ctx, vae, max_batch=8192, loss_weights=weights, lr=1e-3, seed=seed)
tr.push()
perm = torch.randperm(N, device="cuda")


def ref_step(vae, x, step, eps):
    """ae_utils.VAE forward with the trainer's masks and eps, torch autograd."""
    nh = len(hidden)
    h = x
    def block(h, lin, bn, stream):
        z = F.linear(h, lin.weight, lin.bias)
        a = F.leaky_relu(z, 0.01)
        m = torch.from_numpy(keep_mask(seed, step, stream, x.shape[0], lin.out_features, vae.dropout)).to(x.device)
        d = a * m / (1 - vae.dropout)
        mean, var = d.mean(0), d.var(0, unbiased=False)
        return (d - mean) / torch.sqrt(var + 1e-5) * bn.weight + bn.bias
    for i, (lin, bn) in enumerate(zip(vae.encoderlayers, vae.encodernorms)):
        h = block(h, lin, bn, i)
    mu = vae.mu(h)
    ls = F.softplus(vae.logsigma(h))
    z = mu + eps * torch.exp(ls / 2)
    h = z
    for i, (lin, bn) in enumerate(zip(vae.decoderlayers, vae.decodernorms)):
        h = block(h, lin, bn, 50 + i)
    recon = vae.outputlayer(h)
    return vae.calc_loss(x, recon, mu, ls)


# ---- one step, everything compared ------------------------------------------------------
tr.zero_sums()
tr.train(data, perm, B, 1, use_graph=False)
eps = torch.from_numpy(tr.debug(0, B * latent).reshape(B, latent)).cuda()
x = data[perm[:B]]
ref = ae_utils.VAE(cov_size, prof_size, latent_dims=latent, hidden_layers=hidden, device="cuda")
ref.load_state_dict(vae.state_dict())
opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
loss, e_cov, e_comp, kld = ref_step(ref, x, 0, eps)
loss.backward()
sums = tr.sums()
print("loss terms  native", sums, " torch", [float(v) for v in (loss, e_cov, e_comp, kld)])
# gradients: sum of the slice partials
slices = (B + 127) // 128
part = tr.debug(30, slices * tr.n_params).reshape(slices, tr.n_params).sum(0)
ref_tr = NativeTrainer(ctx, ref, max_batch=8192, loss_weights=weights, seed=seed)
grads = []
for t in ref_tr._param_tensors():
    for xx in (t if isinstance(t, tuple) else (t,)):
        grads.append(xx.grad.detach().cpu().numpy().ravel())
g_ref = np.concatenate(grads)
# BatchNorm affine gradients are not in the partials (they come from the backward sums): compare through the update
is_lin = np.ones(tr.n_params, bool)
off = 0
for t in ref_tr._param_tensors():
    for xx in (t if isinstance(t, tuple) else (t,)):
        n = xx.numel()
        if xx.dim() == 1 and any(xx is bn.weight or xx is bn.bias for bn in ref_tr._norms()):
            is_lin[off:off + n] = False
        off += n
err = np.abs(part - g_ref)[is_lin].max() / np.abs(g_ref[is_lin]).max()
print(f"linear-layer gradients: max abs err / max abs = {err:.2e}")
opt.step()
tr.pull()
mx = 0.0
for (k, a), (_, b) in zip(vae.state_dict().items(), ref.state_dict().items()):
    if "num_batches" in k or "running" in k:  # the functional reference above does not keep running statistics
        continue
    d = float((a.float() - b.float()).abs().max())
    mx = max(mx, d)
print(f"after 1 Adam step: max |param - torch| = {mx:.2e} (lr = 1e-3: an Adam step moves every parameter by ~1e-3)")

# ---- 20 more steps, parameters stay together --------------------------------------------
for s in range(1, 21):
    tr.train(data, perm[s * B % (N - B):], B, 1, use_graph=False)
    eps = torch.from_numpy(tr.debug(0, B * latent).reshape(B, latent)).cuda()
    x = data[perm[s * B % (N - B):][:B]]
    opt.zero_grad()
    ref.train()
    l = ref_step(ref, x, s, eps)[0]
    l.backward()
    opt.step()
tr.pull()
mx = max(float((a.float() - b.float()).abs().max()) for (k, a), (_, b) in zip(vae.state_dict().items(), ref.state_dict().items())
         if "num_batches" not in k and "running" not in k)
print(f"after 21 steps: max |param - torch| = {mx:.2e}")

# ---- timing -------------------------------------------------------------------------------
N2 = 432_333
data2 = torch.rand(N2, cov_size + prof_size, device="cuda")
for bs in (1024, 2048, 4096, 8192):
    steps = N2 // bs
    perm2 = torch.randperm(N2, device="cuda")
    tr.train(data2, perm2, bs, steps, use_graph=True); torch.cuda.synchronize()
    t0 = time.time(); tr.train(data2, perm2, bs, steps, use_graph=True); torch.cuda.synchronize(); dt = time.time() - t0
    print(f"native graph  batch {bs}: {dt:.3f} s/epoch, {dt / steps * 1e6:.1f} us/step")
