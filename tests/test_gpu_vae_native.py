"""GPU tests of the fused VAE training step (K7, csrc/lrb_vae.hip) against torch autograd on
the same batch, dropout masks and eps (float32; tolerances stated per check)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(cov_size, prof_size, hidden, latent, n_rows, seed=12345, dropout=None):
    import torch
    from lrbinner_amd import ae_utils, device as lrb
    from lrbinner_amd.vae_native import NativeTrainer
    torch.manual_seed(3)
    rng = np.random.default_rng(7)
    X = rng.random((n_rows, cov_size + prof_size)).astype(np.float32)
    X[:, 1] = 0.0  # a constant column, as MinMax scaling of an empty coverage bin gives
    data = torch.from_numpy(X).cuda()
    vae = ae_utils.VAE(cov_size, prof_size, latent_dims=latent, hidden_layers=hidden, device="cuda")
    if dropout is not None:
        vae.dropout = dropout
    w = ae_utils.h_params.get(str(prof_size), ae_utils.h_params["32"])
    weights = [w["e_cov_weight"], w["e_comp_weight"], w["kld_weight"]]
    ctx = lrb.Context(0, use_torch_stream=True)
    tr = NativeTrainer(ctx, vae, max_batch=4096, loss_weights=weights, lr=1e-3, seed=seed)
    tr.push()
    return torch, ae_utils, vae, data, tr, ctx, weights


def _ref_loss(torch, vae, weights, x, step, eps, seed):
    import torch.nn.functional as F
    from lrbinner_amd.vae_native import keep_mask

    kink = [float("inf")]

    def block(h, lin, bn, stream):
        z = F.linear(h, lin.weight, lin.bias)
        a = F.leaky_relu(z, 0.01)
        m = torch.from_numpy(keep_mask(seed, step, stream, x.shape[0], lin.out_features, vae.dropout)).to(x.device)
        kink[0] = min(kink[0], float(z.detach().abs()[m].min()))
        d = a * m / (1 - vae.dropout)
        mean, var = d.mean(0), d.var(0, unbiased=False)
        return (d - mean) / torch.sqrt(var + 1e-5) * bn.weight + bn.bias, mean, d.var(0, unbiased=True)

    stats = []
    h = x
    for i, (lin, bn) in enumerate(zip(vae.encoderlayers, vae.encodernorms)):
        h, m, v = block(h, lin, bn, i)
        stats.append((m, v))
    mu = vae.mu(h)
    ls = F.softplus(vae.logsigma(h))
    h = mu + eps * torch.exp(ls / 2)
    for i, (lin, bn) in enumerate(zip(vae.decoderlayers, vae.decodernorms)):
        h, m, v = block(h, lin, bn, 50 + i)
        stats.append((m, v))
    recon = vae.outputlayer(h)
    c = vae.cov_size
    diff = (recon - x).pow(2)
    e_cov, e_comp = diff[:, :c].sum(1).mean(), diff[:, c:].sum(1).mean()
    kld = -0.5 * (1 + ls - mu.pow(2) - ls.exp()).sum(1).mean()
    # kink[0]: the smallest kept |pre-activation|.  Below float32 rounding of the dot product its
    # SIGN -- hence LeakyReLU' = 1 or 0.01 -- is decided by the summation order, and the two
    # implementations may legitimately differ in that element's gradient.
    return e_cov * weights[0] + e_comp * weights[1] + kld * weights[2], e_cov, e_comp, kld, stats, kink[0]


@pytest.mark.parametrize("cov_size,prof_size,hidden,latent,B", [(10, 32, [128, 128], 4, 1024), (32, 136, [128, 128], 8, 512),
                                                               (5, 20, [48], 3, 100), (32, 512, [64, 40, 24], 6, 250),
                                                               # beyond the prefetched part of the fused latent layers
                                                               (10, 32, [300, 40], 12, 200), (10, 32, [40, 300], 12, 200),
                                                               # large batches: the batch statistics are summed over 2 / 4 copies
                                                               (10, 32, [128, 128], 4, 2048), (32, 136, [128, 128], 8, 4096)])
def test_steps_match_autograd(cov_size, prof_size, hidden, latent, B):
    seed = 999
    n_rows = max(3000, B + 600)
    torch, ae_utils, vae, data, tr, ctx, weights = _setup(cov_size, prof_size, hidden, latent, n_rows, seed=seed)
    ref = ae_utils.VAE(cov_size, prof_size, latent_dims=latent, hidden_layers=hidden, device="cuda")
    ref.load_state_dict(vae.state_dict())
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    from lrbinner_amd.vae_native import NativeTrainer
    tr_ref = NativeTrainer(ctx, ref, max_batch=4096, loss_weights=weights, seed=seed)  # only for its tensor order
    perm = torch.randperm(n_rows, device="cuda")
    run_mean = [torch.zeros_like(bn.running_mean) for bn in tr._norms()]
    run_var = [torch.ones_like(bn.running_var) for bn in tr._norms()]
    for step in range(4):
        idx = perm[(step * 37) % 500:]
        tr.zero_sums()
        tr.train(data, idx, B, 1, use_graph=False)
        eps = torch.from_numpy(tr.debug(0, B * latent).reshape(B, latent)).cuda()
        opt.zero_grad()
        loss, e_cov, e_comp, kld, stats, kink = _ref_loss(torch, ref, weights, data[idx[:B]], step, eps, seed)
        loss.backward()
        got = tr.sums()
        want = np.array([float(loss.detach()), float(e_cov.detach()), float(e_comp.detach()), float(kld.detach())])
        np.testing.assert_allclose(got, want, rtol=2e-5)
        # every Linear gradient (sum of the batch-slice partials) against autograd
        slices = (B + 127) // 128
        g_native = tr.debug(30, slices * tr.n_params).reshape(slices, tr.n_params).sum(0)
        off = 0
        for t in tr_ref._param_tensors():
            for xx in (t if isinstance(t, tuple) else (t,)):
                n = xx.numel()
                if kink > 1e-5 and not any(xx is bn.weight or xx is bn.bias for bn in tr_ref._norms()):
                    g = xx.grad.detach().cpu().numpy().ravel()
                    assert np.abs(g_native[off:off + n] - g).max() <= 2e-5 * max(np.abs(g).max(), 1e-6), (step, off)
                off += n
        opt.step()
        for q, (m, v) in enumerate(stats):
            run_mean[q] = 0.9 * run_mean[q] + 0.1 * m.detach()
            run_var[q] = 0.9 * run_var[q] + 0.1 * v.detach()
        tr.pull()
        # Adam divides by sqrt(v) + 1e-8: where a gradient is itself ~1e-8 the update (up to lr)
        # depends on its last bits, so a handful of elements may differ by a step; the rest
        # agree to 1e-6.  The reference then continues from the trainer's parameters so that
        # the next step's loss and gradients are compared like for like.
        sd = vae.state_dict()
        for k, b in ref.state_dict().items():
            if "running" in k or "num_batches" in k:
                continue
            d = (sd[k] - b).abs().flatten().float()
            assert float(d.max()) < 2.1e-3, (step, k)
            if kink > 1e-5:
                assert float(torch.quantile(d, 0.99)) < 1e-6 + 1e-5 * float(b.abs().max()), (step, k)
            b.copy_(sd[k])
    for q, bn in enumerate(tr._norms()):
        np.testing.assert_allclose(bn.running_mean.cpu().numpy(), run_mean[q].cpu().numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(bn.running_var.cpu().numpy(), run_var[q].cpu().numpy(), rtol=1e-4, atol=1e-6)
        assert int(bn.num_batches_tracked) == 4
    tr.close()
    tr_ref.close()


def test_graph_replay_equals_plain_launches():
    """The recorded step replayed n times = n plain enqueues (same seed, same permutation)."""
    results = []
    for use_graph in (False, True):
        torch, ae_utils, vae, data, tr, ctx, weights = _setup(10, 32, [128, 128], 4, 5000, seed=5)
        torch.manual_seed(11)
        perm = torch.randperm(5000, device="cuda")
        tr.zero_sums()
        tr.train(data, perm, 1024, 4, use_graph=use_graph)
        tr.train(data, perm, 2048, 2, use_graph=use_graph)
        results.append((tr.get(0, tr.n_params), tr.sums(), tr.steps_done()))
        tr.close()
    assert results[0][2] == results[1][2] == 6
    # float atomics make the batch statistics order-dependent in the last bit; Adam turns the last
    # bit of a ~1e-8 gradient, and LeakyReLU the sign of a ~1e-7 pre-activation, into a visible
    # step for a few elements, which then spreads a little over six steps
    d = np.abs(results[0][0] - results[1][0])
    assert d.max() < 6.1e-3 and np.quantile(d, 0.9) < 5e-5
    np.testing.assert_allclose(results[0][1], results[1][1], rtol=1e-3)


@pytest.mark.parametrize("shape", [(10, 32, 4, 1024, 2048), (32, 136, 8, 2048, 4096)])
def test_deterministic_mode_repeats_bit_for_bit(shape, monkeypatch):
    """LRB_VAE_DETERMINISTIC=1: the batch sums are added up in a fixed order (plain stores per row tile + an ordered
    reduction behind every launch, no float atomics), so the same seed and permutation give the SAME parameters,
    running statistics and loss sums, bit for bit -- run after run, plain launches and graph replays alike, at batch
    sizes on both sides of the 2048-row switch to spread-out sums; and they stay within the atomics' own spread of
    the default mode's."""
    cov, prof, latent, b1, b2 = shape
    rows = 20_000
    runs = {}
    for mode, use_graph in (("1", False), ("1", True), ("1", False), ("0", False)):
        monkeypatch.setenv("LRB_VAE_DETERMINISTIC", mode)
        torch, ae_utils, vae, data, tr, ctx, weights = _setup(cov, prof, [128, 128], latent, rows, seed=5)
        torch.manual_seed(11)
        perm = torch.randperm(rows, device="cuda")
        tr.zero_sums()
        tr.train(data, perm, b1, 3, use_graph=use_graph)
        tr.train(data, perm, b2, 2, use_graph=use_graph)
        tr.pull()
        run = [tr.get(0, tr.n_params).copy(), np.asarray(tr.sums()).copy()] + \
            [bn.running_mean.cpu().numpy().copy() for bn in tr._norms()] + [bn.running_var.cpu().numpy().copy() for bn in tr._norms()]
        runs.setdefault(mode, []).append(run)
        tr.close()
    first = runs["1"][0]
    for other in runs["1"][1:]:
        for a, b in zip(first, other):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    # (the default mode's own runs differ from each other by as much: the last bit of a batch sum, turned into a visible
    # step for a few parameters by Adam and LeakyReLU and spread a little over five steps -- the bulk agrees closely)
    d = np.abs(first[0] - runs["0"][0][0])
    assert np.quantile(d, 0.5) < 2e-5 and np.quantile(d, 0.99) < 2e-3 and d.max() < 5e-2, (np.quantile(d, [0.5, 0.9, 0.99]), d.max())
    np.testing.assert_allclose(first[1], runs["0"][0][1], rtol=2e-3)


def test_native_training_learns_and_keeps_module_contract(tmp_path):
    """trainmodel() on CUDA takes the fused path: the loss falls as it does on the torch path,
    model.pt has the reference's keys, encode() works from the pulled parameters."""
    import torch
    from lrbinner_amd import ae_utils
    rng = np.random.default_rng(0)
    N = 40_000
    centers = rng.random((6, 42))
    prof = centers[rng.integers(0, 6, N)] + rng.normal(size=(N, 42)) * 0.05
    data = ae_utils.make_data(prof[:, :10], prof[:, 10:], "cuda")
    finals = {}
    for native in ("1", "0"):
        import os
        os.environ["LRB_VAE_NATIVE"] = native
        torch.manual_seed(0)
        vae = ae_utils.VAE(10, 32, latent_dims=4, hidden_layers=[128, 128], device="cuda")

        def eval_loss():
            vae.eval()
            with torch.no_grad():
                mu, ls = vae._encode(data[:20000])
                return float(vae.calc_loss(data[:20000], vae._decode(mu), mu, ls)[0])
        start = eval_loss()
        vae.trainmodel(data, nepochs=6, batchsteps=[2, 4], save_path=str(tmp_path / f"m{native}.pt"))
        finals[native] = eval_loss()
        assert finals[native] < 0.5 * start
        assert int(vae.encodernorms[0].num_batches_tracked) == 2 * 39 + 2 * 19 + 2 * 9
        saved = torch.load(str(tmp_path / f"m{native}.pt"), weights_only=False)
        assert set(saved) == {"cov_size", "prof_size", "dropout", "hidden_layers", "latent_dims", "state"}
        assert all(torch.isfinite(v).all() for v in saved["state"].values())
        lat = vae.encode(data)
        assert lat.shape == (N, 4) and lat.dtype == np.float32 and np.isfinite(lat).all()
    os.environ.pop("LRB_VAE_NATIVE", None)
    assert abs(finals["1"] - finals["0"]) < 0.25 * max(finals.values()), finals


def test_argument_errors():
    from lrbinner_amd import _lib
    torch, ae_utils, vae, data, tr, ctx, weights = _setup(10, 32, [128, 128], 4, 2000)
    perm = torch.randperm(2000, device="cuda")
    with pytest.raises(_lib.LrbError):
        tr.train(data, perm, 8192, 1)        # beyond max_batch
    with pytest.raises(_lib.LrbError):
        tr.train(data, perm, 1, 1)           # BatchNorm needs two rows
    with pytest.raises(_lib.LrbError):
        tr.get(9, 4)
    tr.train(data, perm, 1024, 0)            # nothing to do
    assert tr.steps_done() == 0
    tr.close()


def test_native_encode_equals_module_encode():
    """lrb_vae_encode_dev (eval mode: running statistics, no dropout, mu) against the torch module
    with the same parameters; float32 GEMMs in different summation orders: 2e-5 absolute on
    latents of order 1."""
    torch, ae_utils, vae, data, tr, ctx, weights = _setup(32, 136, [128, 128], 8, 20_001)
    perm = torch.randperm(20_001, device="cuda")
    tr.train(data, perm, 1024, 10)          # move the running statistics away from (0, 1)
    tr.pull()
    got = tr.encode(data).cpu().numpy()
    import os
    os.environ["LRB_VAE_NATIVE"] = "0"
    try:
        want = vae.encode(data)
    finally:
        os.environ.pop("LRB_VAE_NATIVE", None)
    assert got.shape == want.shape == (20_001, 8)
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-5 * max(1.0, float(np.abs(want).max())))
    tr.close()


def _load_train_fixture(tag):
    import os
    from helpers import golden_path
    z = np.load(golden_path("py_vae_train.npz"))
    pre = tag + "."
    return z, {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}


@pytest.mark.parametrize("tag,cov_size,prof_size,latent", [("c1", 10, 32, 4), ("c3", 32, 136, 8)])
def test_train_step_against_reference_fixture(tag, cov_size, prof_size, latent):
    """K7 against the REFERENCE's own training step (tests/golden/py_vae_train.npz, written by
    make_golden_py.vae_train_fixture from the imported reference VAE.trainepoch with dropout 0 and a
    pinned eps; ae_utils.py:163-191,199-271): same initial state_dict, same 1024-row batch, same eps
    -> loss terms 2e-5 relative, every Linear gradient 2e-5 of its maximum, parameters after one and
    two Adam steps, BatchNorm running statistics 1e-4 (float32, different summation orders)."""
    import torch
    from lrbinner_amd import ae_utils, device as lrb
    from lrbinner_amd.vae_native import NativeTrainer
    z, fx = _load_train_fixture(tag)
    seed, B = int(z["seed"]), int(z["batch"])
    vae = ae_utils.VAE(cov_size, prof_size, latent_dims=latent, hidden_layers=[128, 128], device="cuda")
    init = {k[len("init."):]: torch.from_numpy(v) for k, v in fx.items() if k.startswith("init.")}
    assert set(init) == set(vae.state_dict())
    vae.load_state_dict(init)
    vae.dropout = 0.0
    w = ae_utils.h_params[str(prof_size)]
    weights = [w["e_cov_weight"], w["e_comp_weight"], w["kld_weight"]]
    ctx = lrb.Context(0, use_torch_stream=True)
    tr = NativeTrainer(ctx, vae, max_batch=B, loss_weights=weights, lr=1e-3, seed=seed)
    tr.push()
    data = torch.from_numpy(fx["X"]).cuda()
    perm = torch.arange(B, device="cuda")
    names = [k for k, _ in vae.named_parameters()]
    for step in range(2):
        tr.zero_sums()
        tr.train(data, perm, B, 1, use_graph=False)
        torch.cuda.synchronize()
        eps = tr.debug(0, B * latent).reshape(B, latent)
        np.testing.assert_allclose(eps, fx[f"s{step}.eps"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(tr.sums(), fx[f"s{step}.loss_terms"], rtol=2e-5)
        if step == 0:
            slices = (B + 127) // 128
            g_native = tr.debug(30, slices * tr.n_params).reshape(slices, tr.n_params).sum(0)
            off = 0
            for t in tr._param_tensors():
                for xx in (t if isinstance(t, tuple) else (t,)):
                    n = xx.numel()
                    name = next(k for k, p in vae.named_parameters() if p is xx)
                    if "norms" not in name:       # BatchNorm affine gradients come from the backward sums
                        g = fx[f"s0.grad.{name}"].ravel()
                        # one LeakyReLU kink (|pre-activation| ~ 1e-7, sign decided by summation order)
                        # moves single elements; everything else agrees to 2e-5 of the maximum
                        d = np.abs(g_native[off:off + n] - g)
                        assert np.quantile(d, 0.999) <= 2e-5 * max(np.abs(g).max(), 1e-6), (name, d.max())
                    off += n
        tr.pull()
        sd = vae.state_dict()
        for k in names + [k for k in sd if "running" in k]:
            want = fx[f"s{step}.post.{k}"]
            got = sd[k].cpu().numpy()
            d = np.abs(got - want).ravel()
            if "running" in k:
                # from the second step on a block's batch MEAN carries the +-lr the noise-gradient biases
                # (below) took in the steps before -- a shift BatchNorm removes again: 0.1 x 2 lr per step
                atol = 1e-6 + (2.5e-4 * step if "running_mean" in k else 0.0)
                np.testing.assert_allclose(got, want, rtol=1e-4, atol=atol, err_msg=k)
            else:
                # Adam's g / (sqrt(v) + 1e-8): where g ~ 1e-8 the update depends on its last bits
                assert d.max() < 2.1e-3 * (step + 1), (step, k, d.max())   # at most one lr-sized step apart per step
                # the bias of a Linear that feeds a BatchNorm has gradient exactly 0 in exact arithmetic
                # (the batch mean is subtracted again); what either implementation holds there is
                # rounding noise of order 1e-9, which Adam normalises to a full +-lr step
                if not (k.endswith("layers.0.bias") or k.endswith("layers.1.bias")):
                    assert np.quantile(d, 0.99) < 1e-6 + 1e-5 * np.abs(want).max(), (step, k, np.quantile(d, 0.99))
        assert int(sd["encodernorms.0.num_batches_tracked"]) == int(fx[f"s{step}.post.encodernorms.0.num_batches_tracked"])
    tr.close()


def test_native_encode_against_reference_latents():
    """lrb_vae_encode_dev straight against the reference's own VAE.encode output
    (tests/golden/py_vae.npz 'latent', make_golden_py.vae_fixture; ae_utils.py:141-161): 1e-4
    absolute on float32 latents (the module test's tolerance)."""
    import torch
    from helpers import golden_path
    from lrbinner_amd import ae_utils, device as lrb
    from lrbinner_amd.vae_native import NativeTrainer
    z = np.load(golden_path("py_vae.npz"))
    hidden = [int(h) for h in z["hidden_layers"]]
    cov_size, prof_size = z["covs_scaled"].shape[1], z["profs_scaled"].shape[1]
    vae = ae_utils.VAE(cov_size, prof_size, latent_dims=z["latent"].shape[1], hidden_layers=hidden, device="cuda")
    vae.load_state_dict({k[len("state."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("state.")})
    ctx = lrb.Context(0, use_torch_stream=True)
    tr = NativeTrainer(ctx, vae, max_batch=1024, loss_weights=[0.1, 1.0, 0.01], seed=1)
    tr.push()
    data = torch.from_numpy(np.concatenate([z["covs_scaled"], z["profs_scaled"]], axis=1).astype(np.float32)).cuda()
    got = tr.encode(data).cpu().numpy()
    assert got.shape == z["latent"].shape
    np.testing.assert_allclose(got, z["latent"], rtol=0, atol=1e-4)
    tr.close()
