"""Host logic of lrbinner_amd.cluster_utils (valley search, cluster peeling, left-over
assignment, output files) against the reference-generated vectors.  The GPU kernels
are replaced here by a test-only numpy backend built on the oracle, so this runs on
CPU; tests/test_gpu_cluster.py runs the same checks through the HIP backend."""
import os
import pickle
import random

import numpy as np
import pytest

from helpers import golden_path
from oracle import np_cluster as oc

from lrbinner_amd import cluster_utils as cu


class OracleBackend:
    """numpy stand-in for HipBackend (tests only)."""

    def load(self, latent):
        self.M = oc.normalize(latent)

    def __len__(self):
        return len(self.M)

    def distances(self, idx):
        return oc.calc_distances(self.M, int(idx))

    def seed_hists(self, seeds):
        return np.stack([oc.histc(oc.calc_distances(self.M, int(s))) for s in seeds]).astype(np.uint32)

    def remove(self, removables):
        keep = np.ones(len(self.M), dtype=bool)
        keep[np.asarray(removables, dtype=np.int64)] = False
        self.M = self.M[keep]


@pytest.fixture(scope="module")
def gc():
    return np.load(golden_path("py_cluster.npz"))


def _nanify(t):
    return [np.nan if (v is False or v is None) else float(v) for v in t]


def test_densities_and_valley_match_reference(gc):
    for h, d in zip(gc["hist"], gc["dens"]):
        assert np.array_equal(cu.calc_densities(h), d)
    for row, exp in zip(gc["fv_in"], gc["fv_out"]):
        got = np.array(_nanify(cu.find_valley_ratio(row)))
        assert np.array_equal(np.isnan(got), np.isnan(exp))
        assert np.array_equal(got[~np.isnan(got)], exp[~np.isnan(exp)])


def test_get_cluster_center_matches_reference(gc):
    b = OracleBackend()
    b.load(gc["latent"])
    for seed_pt, exp in zip((0, 17, 2999), gc["gcc"]):
        random.seed(100 + seed_pt)
        bp, dist, maxima, minima, tail = cu.get_cluster_center(b, seed_pt)
        got = np.array(_nanify((bp, maxima, minima, tail)))
        assert np.array_equal(got, exp, equal_nan=True)
        if bp is not False and bp is not None:
            assert dist[int(bp)] == 0.0


@pytest.mark.parametrize("tag,iters", [("exh", 0), ("it", 40)])
def test_cluster_points_matches_reference(gc, tag, iters):
    random.seed(11)
    clusters = cu.cluster_points(gc["latent"], iters, 500, backend=OracleBackend())
    assert len(clusters) == int(gc[f"cp_{tag}_n"])
    assign = np.full(len(gc["latent"]), -1, dtype=np.int64)
    for order, (cid, members) in enumerate(clusters.items()):
        assign[np.array(sorted(members), dtype=np.int64)] = order
    assert (assign == gc[f"cp_{tag}_assign"]).mean() > 0.999


def test_normal_zero_std_is_nan():
    assert np.isnan(cu.normal(np.array([0.1, 0.2]), np.array([0.1, 0.2]), np.array([0.05, 0.0])))


def _write_case(tmp_path, g):
    out = str(tmp_path)
    os.makedirs(os.path.join(out, "profiles"))
    np.save(os.path.join(out, "latent.npy"), g["latent"])
    np.save(os.path.join(out, "profiles", "com_profs.npy"), g["comp"].astype(np.float64))
    np.save(os.path.join(out, "profiles", "cov_profs.npy"), g["cov"].astype(np.float64))
    reads = os.path.join(out, "reads.fasta")
    with open(reads, "w") as f:
        for i, L in enumerate(g["read_lens"]):
            f.write(f">r{i}\n{'A' * int(L)}\n")
    return out, reads


def test_perform_binning_outputs_match_reference(tmp_path):
    g = np.load(golden_path("py_binning.npz"))
    out, reads = _write_case(tmp_path, g)
    random.seed(21)
    cu.perform_binning(out, 0, 300, True, reads, backend=OracleBackend())
    bins = np.array([int(x) for x in open(os.path.join(out, "bins.txt")).read().split()])
    lengths = np.array([int(x) for x in open(os.path.join(out, "lengths.txt")).read().split()])
    assert np.array_equal(lengths, g["lengths"])
    assert (bins == g["bins"]).mean() > 0.999
    res = pickle.load(open(os.path.join(out, "binning_result.pkl"), "rb"))
    assert sorted(res) == g["result_keys"].tolist()
    assert all(isinstance(v, list) and isinstance(v[0], int) for v in res.values())
    assert [len(res[k]) for k in sorted(res)] == g["result_sizes"].tolist()
    assert sorted(os.listdir(os.path.join(out, "binned_reads"))) == g["binned_files"].tolist()
    first = open(os.path.join(out, "binned_reads", "Bin-0.fasta")).read().split("\n")[:2]
    assert first == g["first_lines"].tolist()


def test_batched_valley_scan_equals_the_scalar_one():
    """find_valley_ratio_batch / calc_densities_batch (vectorised over the sampled seeds) give
    exactly what the per-seed functions give -- on random histograms of every shape the scan
    branches on, and on the golden ones."""
    from lrbinner_amd import cluster_utils as cu
    rng = np.random.default_rng(4)
    hs = []
    for _ in range(400):
        kind = rng.integers(0, 6)
        h = np.zeros(60)
        if kind == 0:
            h = rng.integers(0, 50, 60).astype(float)
        elif kind == 1:      # one peak and a tail
            c = rng.integers(2, 30); h[:] = 200 * np.exp(-0.5 * ((np.arange(60) - c) / rng.uniform(1, 6)) ** 2) + rng.integers(0, 3, 60)
        elif kind == 2:      # two peaks
            for c in rng.integers(0, 60, 2): h += 300 * np.exp(-0.5 * ((np.arange(60) - c) / rng.uniform(1, 4)) ** 2)
        elif kind == 3:      # monotone
            h = np.sort(rng.integers(0, 500, 60))[:: rng.choice([-1, 1])].astype(float)
        elif kind == 4:      # flat / empty
            h[:] = rng.choice([0, 0, 7])
        else:                # plateaus: equal neighbours
            h = np.repeat(rng.integers(0, 100, 12), 5).astype(float)
        hs.append(h)
    H = np.array(hs, dtype=np.float32)
    D = cu.calc_densities_batch(H)
    valid, ratio, maxima, early, minima = cu.find_valley_ratio_batch(D)
    for i in range(len(H)):
        d1 = cu.calc_densities(H[i])
        assert np.array_equal(d1, D[i])
        r = oc.find_valley_ratio(d1)      # the oracle's scalar restatement of cluster_utils.py:87-133
        assert _nanify(cu.find_valley_ratio(d1)) == _nanify(r) or np.array_equal(_nanify(cu.find_valley_ratio(d1)), _nanify(r), equal_nan=True)
        if r[0] is False and r[1] is False:
            assert not valid[i]
            continue
        assert valid[i]
        assert (np.float32(r[0]) == ratio[i]) or (np.isnan(r[0]) and np.isnan(ratio[i]))
        assert r[1] == maxima[i] or (r[1] is None and np.isnan(maxima[i]))
        assert r[2] == early[i] and r[3] == minima[i]


def test_block_valley_scan_equals_the_per_candidate_scan():
    """_valleys_of_hists (one vectorised scan per block of candidates of the exhaustive search) gives, row by row,
    exactly what _valley_of_hist gives -- random histograms, flat ones, empty ones, a single spike."""
    from lrbinner_amd import cluster_utils as cu
    rng = np.random.default_rng(9)
    rows = [rng.poisson(lam, 60) for lam in (0.5, 3, 40, 400) for _ in range(40)]
    rows += [np.zeros(60, np.int64), np.full(60, 7), np.r_[1, np.zeros(59, np.int64)], np.r_[np.zeros(30, np.int64), 5000, np.zeros(29, np.int64)]]
    for _ in range(60):  # peak - valley - second mode, the shape the search is looking for
        x = np.arange(60)
        a, b = rng.integers(2, 12), rng.integers(25, 50)
        rows.append((rng.integers(200, 2000) * np.exp(-0.5 * ((x - a) / rng.uniform(1, 4)) ** 2)
                     + rng.integers(200, 2000) * np.exp(-0.5 * ((x - b) / rng.uniform(2, 6)) ** 2)).astype(np.int64) + 1)
    h = np.stack(rows).astype(np.uint32)
    got = cu._valleys_of_hists(h)
    assert len(got) == len(rows) and cu._valleys_of_hists(h[:0]) == []
    n_found = 0
    for row, g in zip(h, got):
        w = cu._valley_of_hist(row)
        assert len(w) == len(g) == 4
        for a, b in zip(w, g):
            if a is False or b is False or a is None or b is None:
                assert a is b
            else:
                assert (np.isnan(a) and np.isnan(b)) or a == b
        n_found += w[0] is not False
    assert n_found > 50
