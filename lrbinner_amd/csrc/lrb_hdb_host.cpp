// lrb_hdb_host.cpp -- the tree half of HDBSCAN on the host: minimum spanning tree ->
// single-linkage dendrogram -> condensed tree -> cluster stabilities -> excess-of-mass
// selection -> labels.  O(n log n); the O(n^2) distance work is in lrb_hdbscan.hip.
//
// Restates the published algorithm the reference reaches through
// hdbscan.HDBSCAN(min_cluster_size=250).fit_predict (cluster_utils.py:483-495) with that
// package's defaults: cluster_selection_method "eom", allow_single_cluster False,
// cluster_selection_epsilon 0, no max_cluster_size.  (Campello, Moulavi, Sander 2013, sec. 4-5;
// McInnes & Healy 2017 "Accelerated HDBSCAN*", sec. 2.3-2.5.)
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <limits>
#include <numeric>
#include <vector>

#include "lrb_hip.h"
#include "lrb_internal.h"

namespace {

struct cond_row {
    uint32_t parent; // cluster id (>= n)
    uint32_t child;  // point (< n) or cluster id (>= n)
    double lambda;
    uint32_t size;
};

} // namespace

extern "C" int lrb_hdb_labels(uint64_t n64, const uint32_t *eu, const uint32_t *ev, const float *ew,
                              uint32_t min_cluster_size, int32_t *labels, uint32_t *n_clusters)
{
    if (n_clusters) *n_clusters = 0;
    if (n64 == 0) return LRB_OK;
    if (!labels || min_cluster_size < 2 || n64 >= 0x7FFFFFFFull || (n64 > 1 && (!eu || !ev || !ew))) {
        lrb_set_error("invalid argument: lrb_hdb_labels%s%s", "", "");
        return LRB_ERR_ARG;
    }
    const uint32_t n = (uint32_t)n64;
    for (uint32_t i = 0; i < n; ++i) labels[i] = -1;
    if (n < 2) return LRB_OK;
    const uint32_t m = n - 1;

    // ---- single linkage: edges by weight, union-find, one dendrogram node per merge ----
    std::vector<uint32_t> order(m);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return ew[a] < ew[b]; });
    // node ids: 0..n-1 points, n..2n-2 merges (in merge order)
    std::vector<uint32_t> left(m), right(m), size(m);
    std::vector<double> dist(m);
    {
        std::vector<uint32_t> uf(2 * (size_t)n - 1);
        std::iota(uf.begin(), uf.end(), 0u);
        auto find = [&](uint32_t x) {
            uint32_t r = x;
            while (uf[r] != r) r = uf[r];
            while (uf[x] != r) {
                const uint32_t nx = uf[x];
                uf[x] = r;
                x = nx;
            }
            return r;
        };
        for (uint32_t t = 0; t < m; ++t) {
            const uint32_t e = order[t];
            if (eu[e] >= n || ev[e] >= n) {
                lrb_set_error("invalid argument: edge endpoint out of range%s%s", "", "");
                return LRB_ERR_ARG;
            }
            const uint32_t a = find(eu[e]), b = find(ev[e]);
            if (a == b) {
                lrb_set_error("invalid argument: edges do not form a spanning tree%s%s", "", "");
                return LRB_ERR_ARG;
            }
            const uint32_t node = n + t;
            left[t] = a;
            right[t] = b;
            dist[t] = (double)ew[e];
            size[t] = (a < n ? 1u : size[a - n]) + (b < n ? 1u : size[b - n]);
            uf[a] = node;
            uf[b] = node;
        }
    }
    auto node_size = [&](uint32_t x) { return x < n ? 1u : size[x - n]; };

    // ---- condensed tree: walk from the root; a split is real only if both sides hold
    //      min_cluster_size points, otherwise the small side's points fall out of the cluster
    const uint32_t root = 2 * n - 2;
    std::vector<cond_row> rows;
    rows.reserve((size_t)n + n / 4);
    std::vector<uint32_t> relabel(2 * (size_t)n - 1, 0);
    std::vector<char> ignore(2 * (size_t)n - 1, 0);
    uint32_t next_label = n + 1;
    relabel[root] = n;
    std::vector<uint32_t> bfs;
    bfs.reserve(2 * (size_t)n - 1);
    bfs.push_back(root);
    for (size_t h = 0; h < bfs.size(); ++h) { // breadth-first order over the dendrogram
        const uint32_t x = bfs[h];
        if (x >= n) {
            bfs.push_back(left[x - n]);
            bfs.push_back(right[x - n]);
        }
    }
    std::vector<uint32_t> stack;
    auto shed = [&](uint32_t sub, uint32_t parent_label, double lambda) {
        // every point under `sub` leaves parent_label at lambda; the subtree is done
        stack.clear();
        stack.push_back(sub);
        while (!stack.empty()) {
            const uint32_t y = stack.back();
            stack.pop_back();
            ignore[y] = 1;
            if (y < n) {
                rows.push_back({parent_label, y, lambda, 1u});
            } else {
                stack.push_back(right[y - n]);
                stack.push_back(left[y - n]);
            }
        }
    };
    for (size_t h = 0; h < bfs.size(); ++h) {
        const uint32_t x = bfs[h];
        if (ignore[x] || x < n) continue;
        const uint32_t l = left[x - n], r = right[x - n];
        const double d = dist[x - n];
        const double lambda = d > 0.0 ? 1.0 / d : std::numeric_limits<double>::infinity();
        const uint32_t lc = node_size(l), rc = node_size(r);
        const uint32_t pl = relabel[x];
        if (lc >= min_cluster_size && rc >= min_cluster_size) {
            relabel[l] = next_label++;
            rows.push_back({pl, relabel[l], lambda, lc});
            relabel[r] = next_label++;
            rows.push_back({pl, relabel[r], lambda, rc});
        } else if (lc < min_cluster_size && rc < min_cluster_size) {
            shed(l, pl, lambda);
            shed(r, pl, lambda);
        } else if (lc < min_cluster_size) {
            relabel[r] = pl;
            shed(l, pl, lambda);
        } else {
            relabel[l] = pl;
            shed(r, pl, lambda);
        }
    }

    // ---- stability of every cluster: sum over what leaves it of (lambda - lambda_birth)
    const uint32_t n_cl = next_label - n; // cluster ids n .. next_label-1, root = n
    std::vector<double> birth(n_cl, 0.0), stability(n_cl, 0.0);
    std::vector<uint32_t> cparent(n_cl, 0);
    for (const cond_row &rw : rows)
        if (rw.child >= n) {
            birth[rw.child - n] = rw.lambda;
            cparent[rw.child - n] = rw.parent;
        }
    for (const cond_row &rw : rows)
        stability[rw.parent - n] += (rw.lambda - birth[rw.parent - n]) * (double)rw.size;

    // ---- excess of mass: children carry larger ids than their parents, so a descending
    //      sweep sees every child before its parent; the root is never a candidate
    std::vector<char> selected(n_cl, 1);
    selected[0] = 0;
    std::vector<double> child_sum(n_cl, 0.0);
    std::vector<std::vector<uint32_t>> kids(n_cl);
    for (const cond_row &rw : rows)
        if (rw.child >= n) kids[rw.parent - n].push_back(rw.child - n);
    for (uint32_t ci = n_cl; ci-- > 1;) {
        double sub = 0.0;
        for (uint32_t kc : kids[ci]) sub += stability[kc];
        if (sub > stability[ci]) {
            selected[ci] = 0;
            stability[ci] = sub;
        } else {
            // keep this one: everything below it is folded in
            stack.clear();
            for (uint32_t kc : kids[ci]) stack.push_back(kc);
            while (!stack.empty()) {
                const uint32_t y = stack.back();
                stack.pop_back();
                selected[y] = 0;
                for (uint32_t kc : kids[y]) stack.push_back(kc);
            }
        }
    }

    // ---- labels: a point belongs to the nearest selected cluster above where it fell out
    std::vector<int32_t> lab_of(n_cl, -1);
    int32_t next = 0;
    for (uint32_t ci = 1; ci < n_cl; ++ci)
        if (selected[ci]) lab_of[ci] = next++;
    // ascending ids: a parent is resolved before its children
    for (uint32_t ci = 1; ci < n_cl; ++ci)
        if (!selected[ci]) {
            const uint32_t p = cparent[ci] - n;
            lab_of[ci] = p == 0 ? -1 : lab_of[p];
        }
    for (const cond_row &rw : rows)
        if (rw.child < n) labels[rw.child] = rw.parent == n ? -1 : lab_of[rw.parent - n];
    if (n_clusters) *n_clusters = (uint32_t)next;
    return LRB_OK;
}
