"""VAE stage (lrbinner_amd.ae_utils) against vectors produced by the reference's own
ae_utils (tests/golden/py_vae.npz).  CPU here; the same checks run on cuda:0 in
tests/test_gpu_pipeline.py.  Training is stochastic in the reference, so parity is on
the deterministic parts: scaling, encode-with-given-weights, the loss terms, the
saved-model layout."""
import numpy as np
import pytest
import torch

from helpers import golden_path
from lrbinner_amd import ae_utils


@pytest.fixture(scope="module")
def gv():
    return np.load(golden_path("py_vae.npz"))


def load_reference_model(gv, device):
    vae = ae_utils.VAE(10, 32, latent_dims=4, hidden_layers=[32, 32], device=device)
    state = {k[6:]: torch.from_numpy(gv[k]) for k in gv.files if k.startswith("state.")}
    assert set(state) == set(vae.state_dict().keys())  # same state_dict keys as the reference
    vae.load_state_dict(state)
    return vae


def check_encode_and_loss(gv, device, tol):
    vae = load_reference_model(gv, device)
    assert ae_utils.count_parameters(vae) == int(gv["param_count"])
    data = ae_utils.make_data(gv["cov"].astype(np.float64), gv["comp"].astype(np.float64), device)
    ref_scaled = np.concatenate([gv["covs_scaled"], gv["profs_scaled"]], axis=1)
    assert np.array_equal(data.cpu().numpy(), ref_scaled)  # MinMax scaling is bit-exact
    latent = vae.encode(data)
    assert latent.dtype == np.float32 and latent.shape == gv["latent"].shape
    assert np.abs(latent - gv["latent"]).max() < tol
    vae.eval()
    with torch.no_grad():
        x = data[:64]
        mu, logsigma = vae._encode(x)
        recon = torch.from_numpy(np.concatenate([gv["covs_out"], gv["profs_out"]], axis=1)).to(device)
        terms = [float(t) for t in vae.calc_loss(x, recon, mu, logsigma)]
    assert np.allclose(terms, gv["loss_terms"], rtol=1e-4)


def test_encode_and_loss_cpu(gv):
    check_encode_and_loss(gv, "cpu", 1e-5)


def test_saved_model_layout(gv, tmp_path):
    vae = load_reference_model(gv, "cpu")
    p = str(tmp_path / "model.pt")
    vae.save(p)
    saved = torch.load(p, weights_only=False)
    assert sorted(k for k in saved if k != "state") == gv["meta_keys"].tolist()
    assert saved["hidden_layers"] == gv["hidden_layers"].tolist()
    assert set(saved["state"]) == {k[6:] for k in gv.files if k.startswith("state.")}


def test_small_dataset_trains_zero_steps(gv):
    # N < batch with drop_last: the reference silently does nothing (ae_utils.py:19,203-211)
    vae = ae_utils.VAE(10, 32, latent_dims=4, hidden_layers=[8, 8])
    before = {k: v.clone() for k, v in vae.state_dict().items()}
    data = torch.rand(100, 42)
    vae.trainmodel(data, nepochs=2, batchsteps=[])
    assert all(torch.equal(before[k], v) for k, v in vae.state_dict().items())


def test_training_reduces_loss_and_batch_doubles():
    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    cov, comp = rng.random((3000, 10)), rng.random((3000, 32))
    vae = ae_utils.VAE(10, 32, latent_dims=4, hidden_layers=[32, 32])
    data = ae_utils.make_data(cov, comp, "cpu")

    def epoch_loss():
        vae.eval()
        with torch.no_grad():
            mu, ls = vae._encode(data)
            recon = vae._decode(mu)
            return float(vae.calc_loss(data, recon, mu, ls)[0])

    l0 = epoch_loss()
    vae.trainmodel(data, nepochs=6, batchsteps=[2, 4])
    assert epoch_loss() < l0


def test_unknown_profile_width_is_a_key_error():
    vae = ae_utils.VAE(10, 33, latent_dims=2, hidden_layers=[8])
    x = torch.rand(4, 43)
    with pytest.raises(KeyError):
        vae.calc_loss(x, x, torch.zeros(4, 2), torch.zeros(4, 2))
