"""numpy oracle (oracle/np_cluster.py, oracle/np_vae.py) against vectors produced by
importing the reference's own Python (tests/golden/make_golden_py.py).  CPU only."""
import random

import numpy as np
import pytest

from helpers import golden_path
from oracle import np_cluster as oc
from oracle import np_vae as ov


@pytest.fixture(scope="module")
def gc():
    return np.load(golden_path("py_cluster.npz"))


@pytest.fixture(scope="module")
def gv():
    return np.load(golden_path("py_vae.npz"))


def _same_or_nan(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.array_equal(np.isnan(a), np.isnan(b)) and np.allclose(a[~np.isnan(a)], b[~np.isnan(b)], rtol=0, atol=0)


def test_pdf_kernel_is_the_reference_table(gc):
    # one real density row pins the table: dens = hist (*) pdf, bit for bit
    for h, d in zip(gc["hist"], gc["dens"]):
        assert np.array_equal(oc.calc_densities(h), d)


def test_normalize(gc):
    assert np.allclose(oc.normalize(gc["norm_in"]), gc["norm_out"], rtol=0, atol=1e-7)
    assert np.allclose(oc.normalize(gc["norm_in"])[3], 0.25 / (0.5 * 2 ** 0.5), atol=1e-7)


def test_distances_within_tolerance(gc):
    M = oc.normalize(gc["latent"])
    for s, d in zip(gc["seeds"], gc["dist"]):
        got = oc.calc_distances(M, int(s))
        assert np.abs(got - d).max() < 1e-6
        assert got[int(s)] == 0.0


def test_histc_on_reference_distances(gc):
    for d, h in zip(gc["dist"], gc["hist"]):
        mine = oc.histc(d)
        mine[0] -= 1
        assert np.array_equal(mine, h)


def test_histc_edge_probe():
    # SURVEY appendix B, probed on torch: 0.005f falls in bin 0, 0.3 is inclusive
    h = oc.histc(np.array([0, 0.005, 0.2999, 0.3, 0.31, -0.01], dtype=np.float32))
    assert h[0] == 2 and h[59] == 2 and h.sum() == 4


def test_find_valley_ratio(gc):
    for row, exp in zip(gc["fv_in"], gc["fv_out"]):
        got = oc.find_valley_ratio(row)
        got = [np.nan if (v is False or v is None) else float(v) for v in got]
        assert _same_or_nan(got, exp), (row[:12], got, exp)


def test_find_valley_probes():
    # the overwrite quirk gives 90/900, not 90/1000
    r = oc.find_valley_ratio([10, 500, 800, 1000, 900, 500, 100, 90, 95] + [0] * 51)
    assert abs(float(r[0]) - 0.1) < 1e-7 and r[1] == 0.015 and r[2] == 0.030000000000000002 and r[3] == 0.035
    up = list(np.linspace(1, 2000, 21)) + [1500, 900, 100] + [0] * 36
    assert oc.find_valley_ratio(up) == (False, False, False, False)


def test_get_cluster_center(gc):
    M = oc.normalize(gc["latent"])
    for seed_pt, exp in zip((0, 17, 2999), gc["gcc"]):
        random.seed(100 + seed_pt)
        bp, dist, maxima, minima, tail = oc.get_cluster_center(M, seed_pt)
        got = [np.nan if (v is False or v is None) else float(v) for v in (bp, maxima, minima, tail)]
        assert _same_or_nan(got, exp), (got, exp)


@pytest.mark.parametrize("tag,iters", [("exh", 0), ("it", 40)])
def test_cluster_points(gc, tag, iters):
    random.seed(11)
    clusters = oc.cluster_points(gc["latent"], iters, 500)
    assert len(clusters) == int(gc[f"cp_{tag}_n"])
    assign = np.full(len(gc["latent"]), -1, dtype=np.int64)
    for order, (cid, members) in enumerate(clusters.items()):
        assign[np.array(sorted(members), dtype=np.int64)] = order
    assert (assign == gc[f"cp_{tag}_assign"]).mean() > 0.999


def test_minmax_and_encode(gv):
    cov, comp = gv["cov"].astype(np.float64), gv["comp"].astype(np.float64)
    cs, ps = ov.minmax_scale(cov).astype(np.float32), ov.minmax_scale(comp).astype(np.float32)
    assert np.array_equal(cs, gv["covs_scaled"]) and np.array_equal(ps, gv["profs_scaled"])
    assert (cs[:, 7] == 0).all()  # constant column -> zeros, not nan
    state = {k[6:]: gv[k] for k in gv.files if k.startswith("state.")}
    mu, ls = ov.encode(state, cs, ps, n_layers=2)
    assert np.abs(mu - gv["latent"]).max() < 1e-5
    assert np.abs(mu[:64] - gv["mu64"]).max() < 1e-5 and np.abs(ls[:64] - gv["logsigma64"]).max() < 1e-5


def test_loss_terms(gv):
    w = {"kld_weight": 0.00625, "e_cov_weight": 0.1, "e_comp_weight": 1}  # hyper_params "32"
    got = ov.loss_terms(gv["covs_scaled"][:64], gv["covs_out"], gv["profs_scaled"][:64],
                        gv["profs_out"], gv["mu64"], gv["logsigma64"], w)
    assert np.allclose(got, gv["loss_terms"], rtol=1e-5)
