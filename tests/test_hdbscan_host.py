"""Host half of the HDBSCAN path (lrb_hdb_labels): spanning tree -> labels, no GPU.
The spanning trees come from scipy/numpy here; expected labels are those of
sklearn.cluster.HDBSCAN recorded in tests/golden/hdbscan.npz (make_golden_hdbscan.py)."""
import numpy as np
import pytest

from helpers import adjusted_rand, golden_path


@pytest.fixture(scope="module")
def gold():
    return np.load(golden_path("hdbscan.npz"))


def mreach_mst(X, ms):
    """(u, v, w) of the mutual-reachability spanning tree, float64 brute force."""
    from scipy.sparse.csgraph import minimum_spanning_tree
    X = X.astype(np.float64)
    D = np.sqrt(((X[:, None, :] - X[None, :, :]) ** 2).sum(-1))
    core = np.sort(D, axis=1)[:, ms - 1]
    mr = np.maximum(D, np.maximum(core[:, None], core[None, :]))
    np.fill_diagonal(mr, 0)
    t = minimum_spanning_tree(mr).tocoo()
    return t.row.astype(np.uint32), t.col.astype(np.uint32), t.data.astype(np.float32)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_labels_from_spanning_tree_match_sklearn(gold, tag):
    from lrbinner_amd import device as lrb
    X = gold[f"{tag}_X"]
    _, _, _, _, mcs, ms = (int(v) for v in gold[f"{tag}_params"])
    u, v, w = mreach_mst(X, ms)
    assert len(u) == len(X) - 1
    labels, nc = lrb.hdb_labels(len(X), u, v, w, mcs)
    ref = gold[f"{tag}_labels"]
    assert nc == ref.max() + 1
    assert adjusted_rand(labels, ref) >= 0.99
    assert abs(int((labels < 0).sum()) - int((ref < 0).sum())) <= max(3, len(X) // 500)


def test_labels_hand_case():
    """Two chains of 6 points 1 apart, 10 apart from each other, one far outlier; mcs 4:
    both chains are clusters, the outlier is noise."""
    from lrbinner_amd import device as lrb
    n = 13
    u = list(range(0, 5)) + list(range(6, 11)) + [5, 11]
    v = list(range(1, 6)) + list(range(7, 12)) + [6, 12]
    w = [1.0] * 10 + [10.0, 50.0]
    labels, nc = lrb.hdb_labels(n, u, v, w, 4)
    assert nc == 2
    assert len(set(labels[:6])) == 1 and len(set(labels[6:12])) == 1
    assert labels[0] != labels[6] and labels[0] >= 0 and labels[6] >= 0
    assert labels[12] == -1


def test_labels_single_blob_is_all_noise():
    """allow_single_cluster is False in the package's defaults: one blob -> no cluster."""
    from lrbinner_amd import device as lrb
    rng = np.random.default_rng(5)
    X = rng.normal(size=(400, 3)).astype(np.float32)
    u, v, w = mreach_mst(X, 10)
    labels, nc = lrb.hdb_labels(len(X), u, v, w, 200)
    assert nc == 0 and (labels == -1).all()


def test_labels_degenerate_inputs():
    from lrbinner_amd import device as lrb, _lib
    labels, nc = lrb.hdb_labels(0, [], [], [], 5)
    assert len(labels) == 0 and nc == 0
    labels, nc = lrb.hdb_labels(1, [], [], [], 5)
    assert labels.tolist() == [-1]
    # zero-weight edges (duplicate points): lambda = inf, must not crash
    labels, nc = lrb.hdb_labels(6, [0, 1, 2, 3, 4], [1, 2, 3, 4, 5], [0, 0, 0, 0, 0], 2)
    assert len(labels) == 6
    # not a spanning tree / endpoint out of range / min_cluster_size < 2
    with pytest.raises(_lib.LrbError):
        lrb.hdb_labels(4, [0, 0, 2], [1, 1, 3], [1, 1, 1], 2)
    with pytest.raises(_lib.LrbError):
        lrb.hdb_labels(3, [0, 1], [1, 7], [1, 1], 2)
    with pytest.raises(_lib.LrbError):
        lrb.hdb_labels(3, [0, 1], [1, 2], [1, 1], 1)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_oracle_restatement_matches_sklearn_vectors(gold, tag):
    """oracle/np_hdbscan.py (the CPU restatement this row is checked with) against the vectors of
    the available implementation: core distances exactly, the spanning-tree weight, the labels."""
    from oracle import np_hdbscan as oh
    X = gold[f"{tag}_X"]
    _, _, _, _, mcs, ms = (int(v) for v in gold[f"{tag}_params"])
    W, core = oh.mutual_reachability(X, ms)
    np.testing.assert_allclose(core, gold[f"{tag}_core"], rtol=1e-12)
    np.fill_diagonal(W, np.inf)
    u, v, w = oh.mst_prim(W)
    assert abs(w.sum() / float(gold[f"{tag}_mst_weight"][0]) - 1) < 1e-9
    labels, nc = oh.labels_from_mst(len(X), u, v, w, mcs)
    ref = gold[f"{tag}_labels"]
    assert nc == ref.max() + 1 and adjusted_rand(labels, ref) >= 0.995  # ties: sklearn works on the float32 input


@pytest.mark.parametrize("seed,n,d,mcs,ms", [(1, 600, 2, 15, 5), (2, 900, 5, 40, 40), (3, 400, 3, 10, 3), (4, 700, 8, 25, 10)])
def test_library_labels_equal_the_oracle(seed, n, d, mcs, ms):
    """lrb_hdb_labels against the oracle on the oracle's own spanning tree: same partition
    (float32 edge weights in the library: ties in float32 may order two merges differently)."""
    from lrbinner_amd import device as lrb
    from oracle import np_hdbscan as oh
    rng = np.random.default_rng(seed)
    cents = rng.normal(size=(4, d)) * 4
    X = np.concatenate([c + rng.normal(size=(n // 4, d)) * rng.uniform(0.3, 1.2) for c in cents] +
                       [rng.uniform(-10, 10, size=(n // 10, d))])
    W, _ = oh.mutual_reachability(X, ms)
    np.fill_diagonal(W, np.inf)
    u, v, w = oh.mst_prim(W)
    want, nc_want = oh.labels_from_mst(len(X), u, v, w, mcs)
    got, nc = lrb.hdb_labels(len(X), u, v, w, mcs)
    assert nc == nc_want
    assert adjusted_rand(got, want) >= 0.995
