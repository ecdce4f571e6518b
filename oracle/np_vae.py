"""numpy restatement of the VAE's deterministic arithmetic -- TEST INFRASTRUCTURE.

Follows ``mbcclr_utils/ae_utils.py``; pinned against vectors produced by the
reference itself (tests/golden/py_vae.npz).  float32 throughout; agreement with
torch is to rounding (different GEMM summation order), tolerance 1e-5.
"""
import numpy as np


def minmax_scale(x):
    """MinMaxScaler().fit_transform per column (ae_utils.py:21-22): x*scale + min_,
    a constant column maps to 0."""
    x = np.asarray(x, dtype=np.float64)
    lo, hi = x.min(axis=0), x.max(axis=0)
    rng = hi - lo
    rng[rng == 0] = 1.0
    scale = 1.0 / rng
    return x * scale + (0.0 - lo * scale)


def _linear(x, w, b):
    return (x @ w.T + b).astype(np.float32)


def _block(x, state, prefix_layer, prefix_norm, i):
    """BatchNorm(Dropout(LeakyReLU(Linear(x)))) in eval mode (ae_utils.py:130-133)."""
    h = _linear(x, state[f"{prefix_layer}.{i}.weight"], state[f"{prefix_layer}.{i}.bias"])
    h = np.where(h >= 0, h, h * np.float32(0.01)).astype(np.float32)
    mean, var = state[f"{prefix_norm}.{i}.running_mean"], state[f"{prefix_norm}.{i}.running_var"]
    g, beta = state[f"{prefix_norm}.{i}.weight"], state[f"{prefix_norm}.{i}.bias"]
    return ((h - mean) / np.sqrt(var + np.float32(1e-5)) * g + beta).astype(np.float32)


def encode(state, covs_scaled, profs_scaled, n_layers):
    """(mu, logsigma) of forward_predict (ae_utils.py:127-139,193-197)."""
    x = np.concatenate([covs_scaled, profs_scaled], axis=1).astype(np.float32)
    for i in range(n_layers):
        x = _block(x, state, "encoderlayers", "encodernorms", i)
    mu = _linear(x, state["mu.weight"], state["mu.bias"])
    pre = _linear(x, state["logsigma.weight"], state["logsigma.bias"])
    # Softplus(beta=1, threshold=20)
    logsigma = np.where(pre > 20, pre, np.log1p(np.exp(np.minimum(pre, 20)))).astype(np.float32)
    return mu, logsigma


def loss_terms(cov_in, cov_out, prof_in, prof_out, mu, logsigma, w):
    """calc_loss without constraints (ae_utils.py:255-269): (loss, e_cov, e_comp, kld)."""
    e_cov = np.mean(np.sum((cov_out - cov_in) ** 2, axis=1))
    e_comp = np.mean(np.sum((prof_out - prof_in) ** 2, axis=1))
    kld = -0.5 * np.mean(np.sum(1 + logsigma - mu ** 2 - np.exp(logsigma), axis=1))
    loss = e_cov * w["e_cov_weight"] + e_comp * w["e_comp_weight"] + kld * w["kld_weight"]
    return loss, e_cov, e_comp, kld
