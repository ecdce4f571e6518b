"""GPU tests of the HDBSCAN kernels (K6): exact core distances, the mutual-reachability
spanning tree and the labels against the vectors in tests/golden/hdbscan.npz (float64 brute
force / scipy / sklearn.cluster.HDBSCAN; the reference's own `hdbscan` package is absent and
unpinned -- SURVEY.md 8c -- so this row's parity is against the available implementation)."""
import numpy as np
import pytest

from helpers import adjusted_rand, golden_path

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gold():
    return np.load(golden_path("hdbscan.npz"))


@pytest.fixture(scope="module")
def ctx():
    from lrbinner_amd import device as lrb
    return lrb.Context(0, use_torch_stream=True)


def _spanning(n, u, v):
    parent = list(range(n))

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x
    for a, b in zip(u.tolist(), v.tolist()):
        ra, rb = find(a), find(b)
        if ra == rb:
            return False
        parent[ra] = rb
    return len(u) == n - 1


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_core_mst_labels_match_golden(ctx, gold, tag):
    import torch
    from lrbinner_amd import device as lrb
    X = gold[f"{tag}_X"]
    _, _, _, _, mcs, ms = (int(v) for v in gold[f"{tag}_params"])
    Xt = torch.from_numpy(X).cuda()
    core = ctx.hdb_core_dist_dev(Xt, ms)
    # exact selection: only float32 rounding of the squared distance separates the two
    np.testing.assert_allclose(core.cpu().numpy(), gold[f"{tag}_core"], rtol=2e-6, atol=1e-6)
    u, v, w, rounds = ctx.hdb_mst_dev(Xt, core)
    assert _spanning(len(X), u, v) and rounds >= 1
    assert abs(float(w.astype(np.float64).sum()) / float(gold[f"{tag}_mst_weight"][0]) - 1) < 1e-5
    labels, nc = lrb.hdb_labels(len(X), u, v, w, mcs)
    ref = gold[f"{tag}_labels"]
    assert nc == ref.max() + 1
    assert adjusted_rand(labels, ref) >= 0.99
    # the one-call host entry gives the same labels
    assert np.array_equal(ctx.hdbscan(X, mcs, ms), labels)


@pytest.mark.parametrize("n,dims,k", [(1, 3, 1), (2, 1, 2), (63, 5, 7), (65, 64, 64), (257, 17, 30), (1000, 2, 1)])
def test_core_distances_odd_shapes(ctx, n, dims, k):
    import torch
    rng = np.random.default_rng(n * 100 + dims)
    X = rng.normal(size=(n, dims)).astype(np.float32)
    X64 = X.astype(np.float64)
    D = np.sqrt(((X64[:, None, :] - X64[None, :, :]) ** 2).sum(-1))
    want = np.sort(D, axis=1)[:, k - 1]
    got = ctx.hdb_core_dist_dev(torch.from_numpy(X).cuda(), k).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=3e-6, atol=1e-6)


def test_duplicates_and_argument_errors(ctx):
    import torch
    from lrbinner_amd import _lib
    # 300 copies of each of two points: core distance 0, zero-weight tree edges
    X = np.repeat(np.array([[0, 0, 0], [5, 5, 5]], np.float32), 300, axis=0)
    labels = ctx.hdbscan(X, 100, 100)
    assert len(labels) == 600 and set(labels[:300].tolist()) != set(labels[300:].tolist()) or (labels == -1).all()
    Xt = torch.from_numpy(X).cuda()
    assert float(ctx.hdb_core_dist_dev(Xt, 100).abs().max()) == 0.0
    with pytest.raises(_lib.LrbError):
        ctx.hdb_core_dist_dev(Xt, 601)          # k > n
    with pytest.raises(_lib.LrbError):
        ctx.hdb_core_dist_dev(Xt, 0)
    with pytest.raises(_lib.LrbError):
        ctx.hdbscan(np.zeros((10, 65), np.float32), 5)   # dims > 64
    with pytest.raises(_lib.LrbError):
        ctx.hdbscan(np.zeros((10, 3), np.float32), 20)   # min_samples > n


def test_full_size_properties(ctx):
    """C5-sized input (500 k fragments x 8): the spanning tree spans, weights are
    non-decreasing along no path shorter than a core distance, labels partition the blobs."""
    import torch
    rng = np.random.default_rng(99)
    n, d, k = 500_000, 8, 8
    cents = rng.normal(size=(k, d)) * 4
    truth = rng.integers(0, k, n)
    X = (cents[truth] + rng.normal(size=(n, d)) * 0.5).astype(np.float32)
    Xt = torch.from_numpy(X).cuda()
    core = ctx.hdb_core_dist_dev(Xt, 250)
    u, v, w, rounds = ctx.hdb_mst_dev(Xt, core)
    assert _spanning(n, u, v)
    c = core.cpu().numpy()
    # every edge is at least as long as both endpoints' core distances and the points' distance
    d_uv = np.sqrt(((X[u].astype(np.float64) - X[v].astype(np.float64)) ** 2).sum(1))
    lower = np.maximum(np.maximum(c[u], c[v]), d_uv)
    np.testing.assert_allclose(w, lower, rtol=1e-5, atol=1e-6)
    from lrbinner_amd import device as lrb
    labels, nc = lrb.hdb_labels(n, u, v, w, 250)
    assert nc == k
    keep = labels >= 0
    assert keep.mean() > 0.9 and adjusted_rand(labels[keep], truth[keep]) > 0.99


def _blobs(rng, n, d, n_blobs, noise=0.05):
    centers = rng.normal(size=(n_blobs, d)) * 3.0
    x = centers[rng.integers(0, n_blobs, n)] + rng.normal(size=(n, d)) * rng.uniform(0.05, 0.4, size=(n, 1))
    m = int(n * noise)
    x[:m] = rng.uniform(-8, 8, size=(m, d))
    return x.astype(np.float32)


@pytest.mark.parametrize("kind,n,d,k", [("blobs", 70_001, 8, 250), ("blobs", 65_000, 4, 100), ("uniform", 61_000, 8, 64),
                                          ("duplicates", 66_666, 3, 250), ("blobs", 90_000, 12, 17)])
def test_pruned_core_distances_equal_brute_force(ctx, kind, n, d, k, monkeypatch):
    """The spatially pruned select (Morton order, box lower bounds, window upper bounds) skips only
    tiles that cannot hold a k-th neighbour: its core distances are the brute-force kernel's, bit for
    bit -- clustered data, uniform data (little to prune), many identical points, a row count that
    fills neither the last group nor the last tile, more than 8 dimensions (the order uses 8)."""
    import torch
    rng = np.random.default_rng(n + d)
    if kind == "blobs":
        x = _blobs(rng, n, d, 23)
    elif kind == "uniform":
        x = rng.random((n, d)).astype(np.float32)
    else:
        base = _blobs(rng, 500, d, 7)
        x = base[rng.integers(0, 500, n)]
    xt = torch.from_numpy(x).cuda()
    monkeypatch.setenv("LRB_HDB_BRUTE", "1")
    want = ctx.hdb_core_dist_dev(xt, k).cpu().numpy()
    monkeypatch.setenv("LRB_HDB_BRUTE", "0")
    for window in ("8", "0", "2"):
        monkeypatch.setenv("LRB_HDB_WINDOW", window)
        got = ctx.hdb_core_dist_dev(xt, k).cpu().numpy()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (kind, window)


@pytest.mark.parametrize("kind,n,d,k", [("blobs", 70_001, 8, 250), ("blobs", 64_000, 4, 50), ("uniform", 61_000, 6, 32),
                                          ("duplicates", 66_666, 3, 100)])
def test_pruned_boruvka_steps_give_the_same_tree(ctx, kind, n, d, k, monkeypatch):
    """The spanning tree from the pruned Boruvka steps (Morton order; tiles skipped by component and by
    box distance against the queries' current best edges) is the brute-force one edge for edge: same
    endpoints, same weights, same order -- ties included (duplicates give thousands of equal weights)."""
    import torch
    rng = np.random.default_rng(7 * n + d)
    if kind == "blobs":
        x = _blobs(rng, n, d, 19)
    elif kind == "uniform":
        x = rng.random((n, d)).astype(np.float32)
    else:
        base = _blobs(rng, 700, d, 5)
        x = base[rng.integers(0, 700, n)]
    xt = torch.from_numpy(x).cuda()
    monkeypatch.setenv("LRB_HDB_BRUTE", "1")
    core = ctx.hdb_core_dist_dev(xt, k)
    u0, v0, w0, r0 = ctx.hdb_mst_dev(xt, core)
    monkeypatch.setenv("LRB_HDB_BRUTE", "0")
    for window in ("8", "0"):
        monkeypatch.setenv("LRB_HDB_WINDOW", window)
        u1, v1, w1, r1 = ctx.hdb_mst_dev(xt, core)
        assert r1 == r0
        assert np.array_equal(u1, u0) and np.array_equal(v1, v0), (kind, window)
        assert np.array_equal(w1.view(np.uint32), w0.view(np.uint32))
    assert _spanning(n, u0, v0)
