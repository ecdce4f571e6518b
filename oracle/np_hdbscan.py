"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of HDBSCAN* for the contigs pipeline (SURVEY.md 8a row 28:
``HDBSCAN(min_cluster_size=250).fit_predict(latent)``, cluster_utils.py:483-495).  The
reference calls the third-party ``hdbscan`` package, which is absent from /root/reference
and from this image and is not version-pinned by the reference: PARITY UNPINNED for this
row.  What is restated is the published algorithm with that package's defaults

    Campello, Moulavi, Sander (2013) "Density-based clustering based on hierarchical density
    estimates", sec. 3-5;  McInnes, Healy, Astels (2017) "hdbscan: Hierarchical density
    based clustering" / McInnes & Healy (2017) "Accelerated HDBSCAN*", sec. 2

(euclidean metric, min_samples = min_cluster_size counted with the point itself, alpha 1,
excess-of-mass selection, no single cluster), and it is anchored on the implementation that
IS available offline, sklearn.cluster.HDBSCAN 1.7.2 (tests/golden/hdbscan.npz,
tests/test_hdbscan_host.py).  float64 throughout; O(n^2) memory: small inputs only.
"""
import numpy as np


def pairwise(X):
    X = np.asarray(X, dtype=np.float64)
    return np.sqrt(((X[:, None, :] - X[None, :, :]) ** 2).sum(-1))


def core_distances(X, k):
    """Distance to the k-th nearest row, the row itself included."""
    return np.sort(pairwise(X), axis=1)[:, k - 1]


def mutual_reachability(X, k):
    D = pairwise(X)
    core = np.sort(D, axis=1)[:, k - 1]
    return np.maximum(D, np.maximum(core[:, None], core[None, :])), core


def mst_prim(W):
    """Minimum spanning tree of a dense symmetric weight matrix (Prim).  Returns (u, v, w)."""
    n = len(W)
    in_tree = np.zeros(n, bool)
    best = np.full(n, np.inf)
    frm = np.zeros(n, np.int64)
    in_tree[0] = True
    best[:] = W[0]
    best[0] = np.inf
    u, v, w = [], [], []
    for _ in range(n - 1):
        j = int(np.argmin(np.where(in_tree, np.inf, best)))
        u.append(int(frm[j])); v.append(j); w.append(float(best[j]))
        in_tree[j] = True
        closer = (W[j] < best) & ~in_tree
        best[closer] = W[j][closer]
        frm[closer] = j
    return np.array(u), np.array(v), np.array(w)


def labels_from_mst(n, u, v, w, min_cluster_size):
    """Single linkage -> condensed tree -> stabilities -> excess of mass -> labels (-1 noise)."""
    order = np.argsort(w, kind="stable")
    parent = list(range(2 * n - 1))

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    left, right, dist, size = {}, {}, {}, {i: 1 for i in range(n)}
    for t, e in enumerate(order):
        a, b = find(int(u[e])), find(int(v[e]))
        node = n + t
        left[node], right[node], dist[node] = a, b, float(w[e])
        size[node] = size[a] + size[b]
        parent[a] = parent[b] = node
    root = 2 * n - 2

    def leaves(x):
        stack, out = [x], []
        while stack:
            y = stack.pop()
            if y < n:
                out.append(y)
            else:
                stack += [right[y], left[y]]
        return out

    rows = []  # (parent cluster, child, lambda, size)
    relabel = {root: n}
    next_label = n + 1
    queue = [root]
    while queue:
        x = queue.pop(0)
        if x < n:
            continue
        l, r = left[x], right[x]
        lam = 1.0 / dist[x] if dist[x] > 0 else np.inf
        big_l, big_r = size[l] >= min_cluster_size, size[r] >= min_cluster_size
        if big_l and big_r:
            for c in (l, r):
                relabel[c] = next_label
                rows.append((relabel[x], next_label, lam, size[c]))
                next_label += 1
                queue.append(c)
        else:
            for c, big in ((l, big_l), (r, big_r)):
                if big:
                    relabel[c] = relabel[x]
                    queue.append(c)
                else:
                    rows += [(relabel[x], p, lam, 1) for p in leaves(c)]
    birth = {n: 0.0}
    kids = {}
    for p, c, lam, _ in rows:
        if c >= n:
            birth[c] = lam
            kids.setdefault(p, []).append(c)
    stability = {c: 0.0 for c in birth}
    for p, c, lam, s in rows:
        stability[p] += (lam - birth[p]) * s
    selected = {c: c != n for c in birth}
    for c in sorted(birth, reverse=True):
        if c == n:
            continue
        sub = sum(stability[k] for k in kids.get(c, []))
        if sub > stability[c]:
            selected[c] = False
            stability[c] = sub
        else:
            stack = list(kids.get(c, []))
            while stack:
                y = stack.pop()
                selected[y] = False
                stack += kids.get(y, [])
    cparent = {c: p for p, c, _, _ in rows if c >= n}
    lab, nxt = {}, 0
    for c in sorted(birth):
        if c != n and selected[c]:
            lab[c] = nxt
            nxt += 1
    for c in sorted(birth):
        if c != n and not selected[c]:
            lab[c] = lab.get(cparent[c], -1) if cparent[c] != n else -1
    labels = np.full(n, -1, np.int64)
    for p, c, _, _ in rows:
        if c < n:
            labels[c] = -1 if p == n else lab[p]
    return labels, nxt


def hdbscan(X, min_cluster_size, min_samples=None):
    k = min_cluster_size if min_samples is None else min_samples
    W, _ = mutual_reachability(X, k)
    np.fill_diagonal(W, np.inf)
    u, v, w = mst_prim(W)
    return labels_from_mst(len(X), u, v, w, min_cluster_size)[0]
