// lrb_hdbscan.hip -- the distance work of HDBSCAN for the contigs pipeline
// (cluster_utils.py:483-495: HDBSCAN(min_cluster_size=250).fit_predict(latent)), gfx950 only.
//
// The reference hands the fragment latents (F x latent_dims float32, F up to ~10^6,
// latent_dims 4..64) to the third-party `hdbscan` package.  Its arithmetic is the published
// HDBSCAN* algorithm (Campello, Moulavi, Sander 2013; McInnes, Healy 2017):
//   core_k(x)   = distance from x to its k-th nearest neighbour, x itself included
//   d_mreach    = max(core_k(a), core_k(b), |a - b|)
//   MST of the complete graph under d_mreach -> single linkage -> condensed tree -> EOM
// The two O(F^2) steps run here as brute-force HIP kernels (the data are low-dimensional and
// fit in L2, so an exact all-pairs sweep beats a tree walk on this machine); the O(F log F)
// tree steps are host C++ (lrb_hdb_host.cpp).
//
// Both kernels share one shape: a workgroup of 4 waves owns 64 query points, lane l of every
// wave holds query l in registers; candidates are staged through LDS in tiles of 256 rows,
// wave w sweeps rows w*64.. of the tile, every lane reading the SAME candidate (LDS
// broadcast, no bank conflicts), so the inner loop is 2 VALU per dimension and pair.
//
//   hdb_core_kernel     exact k-th smallest squared distance per query by a 4-pass radix
//                       select over the float bit pattern (non-negative floats order like
//                       their bits): an LDS histogram hist[digit][lane] -- the lane index is
//                       the bank, so the 64 tallies of a wave never collide.
//   hdb_nearest_kernel  one Boruvka step: per point the lightest d_mreach edge into another
//                       component, ties broken by (weight, lower index, higher index) -- a
//                       total order on edges, which keeps Boruvka cycle-free.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>

#include "lrb_device.h"

#define HDB_Q 64     // queries per workgroup (= lanes)
#define HDB_TILE 256 // candidate rows per LDS tile

// rows zero-padded to DP floats (padding adds nothing to a squared difference)
__global__ __launch_bounds__(256) void hdb_pad_kernel(const float *__restrict__ X, uint64_t n, uint32_t dims,
                                                      uint32_t dp, float *__restrict__ Xp)
{
    const uint64_t total = n * dp;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256) {
        const uint64_t r = i / dp;
        const uint32_t d = (uint32_t)(i - r * dp);
        Xp[i] = d < dims ? X[r * dims + d] : 0.0f;
    }
}

template <int DP>
__device__ __forceinline__ void hdb_load_query(const float *__restrict__ Xp, uint32_t row, float (&xq)[DP])
{
    const float4 *src = reinterpret_cast<const float4 *>(Xp + (uint64_t)row * DP);
#pragma unroll
    for (int v = 0; v < DP / 4; ++v) {
        const float4 t = src[v];
        xq[4 * v] = t.x;
        xq[4 * v + 1] = t.y;
        xq[4 * v + 2] = t.z;
        xq[4 * v + 3] = t.w;
    }
}

template <int DP>
__device__ __forceinline__ void hdb_stage_tile(const float *__restrict__ Xp, uint32_t n, uint32_t tile0,
                                               float *tile, uint32_t tid)
{
    // thread t brings row tile0+t (rows past the end are never swept)
    const uint32_t j = tile0 + tid;
    if (j < n) {
        const float4 *src = reinterpret_cast<const float4 *>(Xp + (uint64_t)j * DP);
        float4 *dst = reinterpret_cast<float4 *>(tile + tid * DP);
#pragma unroll
        for (int v = 0; v < DP / 4; ++v) dst[v] = src[v];
    }
}

template <int DP>
__device__ __forceinline__ float hdb_dist2(const float (&xq)[DP], const float *row)
{
    const float4 *r4 = reinterpret_cast<const float4 *>(row);
    float acc = 0.0f;
#pragma unroll
    for (int v = 0; v < DP / 4; ++v) {
        const float4 t = r4[v];
        const float a = xq[4 * v] - t.x, b = xq[4 * v + 1] - t.y, c = xq[4 * v + 2] - t.z,
                    d = xq[4 * v + 3] - t.w;
        acc = fmaf(a, a, acc);
        acc = fmaf(b, b, acc);
        acc = fmaf(c, c, acc);
        acc = fmaf(d, d, acc);
    }
    return acc;
}

template <int DP>
__global__ __launch_bounds__(256) void hdb_core_kernel(const float *__restrict__ Xp, uint32_t n, uint32_t k,
                                                       float *__restrict__ core)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *hist = smem;                                         // [256 digits][64 lanes]
    float *tile = reinterpret_cast<float *>(smem + 256 * HDB_Q);   // [256 rows][DP]
    __shared__ uint32_t s_prefix[HDB_Q], s_krem[HDB_Q];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t q = blockIdx.x * HDB_Q + lane;
    float xq[DP];
    hdb_load_query<DP>(Xp, q < n ? q : n - 1, xq);
    uint32_t prefix = 0, krem = k;
    // 31 key bits (the sign of a squared distance is 0): digits of 8, 8, 8, 7 bits
#pragma unroll 1
    for (int pass = 0; pass < 4; ++pass) {
        const uint32_t width = pass == 3 ? 7u : 8u;
        const uint32_t shift = pass == 0 ? 23u : pass == 1 ? 15u : pass == 2 ? 7u : 0u;
        const uint32_t hs = shift + width; // bits above the digit, already decided
        const uint32_t wmask = (1u << width) - 1u;
        for (uint32_t i = tid; i < 256 * HDB_Q; i += 256) hist[i] = 0;
#pragma unroll 1
        for (uint32_t tile0 = 0; tile0 < n; tile0 += HDB_TILE) {
            lrb_barrier();
            hdb_stage_tile<DP>(Xp, n, tile0, tile, tid);
            lrb_barrier();
            const uint32_t c0 = wave * 64u;
            const uint32_t left = n - tile0;
            const uint32_t cend = left < c0 + 64u ? (left > c0 ? left : c0) : c0 + 64u;
#pragma unroll 4
            for (uint32_t c = c0; c < cend; ++c) {
                const uint32_t key = __float_as_uint(hdb_dist2<DP>(xq, tile + c * DP));
                if ((hs >= 31u ? 0u : key >> hs) == prefix)
                    atomicAdd(&hist[((key >> shift) & wmask) * HDB_Q + lane], 1u);
            }
        }
        lrb_barrier();
        if (wave == 0) {
            // lane l walks the digits of query l: the digit where the running count reaches
            // the remaining rank
            uint32_t cum = 0, digit = wmask;
            for (uint32_t b = 0; b <= wmask; ++b) {
                const uint32_t cnt = hist[b * HDB_Q + lane];
                if (cum + cnt >= krem) {
                    digit = b;
                    break;
                }
                cum += cnt;
            }
            s_prefix[lane] = (prefix << width) | digit;
            s_krem[lane] = krem - cum;
        }
        lrb_barrier();
        prefix = s_prefix[lane];
        krem = s_krem[lane];
        lrb_barrier();
    }
    if (wave == 0 && q < n) core[q] = sqrtf(__uint_as_float(prefix));
}

// ---------------------------------------------------------------------------
// The same select, spatially pruned (same arithmetic per pair, same result bit for bit).
//
// The points are put in Morton order of their coordinates (quantised to 64 / min(dims, 8) bits each, 16 at most), so that the
// 64 queries of a workgroup and the 256 rows of a candidate tile are each a small box in space.  A first select
// over a WINDOW of tiles around the queries' own gives every group an upper bound U on its k-th neighbour
// distances (the k-th of a subset is never below the k-th of the whole); the full select then skips every tile
// whose box is farther from the group's box than U: all its candidates lie beyond every query's k-th neighbour and
// fall into digits above the ones the select is looking for.  On clustered latents that leaves a few per cent of
// the n^2 pairs.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t hdb_ord_bits(float x) // float -> uint, order preserving
{
    const uint32_t b = __float_as_uint(x);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float hdb_ord_float(uint32_t u)
{
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

// mm[d] = min, mm[64 + d] = max of coordinate d, as order-preserving uints (initialised to ~0 / 0)
__global__ __launch_bounds__(256) void hdb_minmax_kernel(const float *__restrict__ Xp, uint32_t n, uint32_t dp, uint32_t nd,
                                                         uint32_t *__restrict__ mm)
{
    __shared__ uint32_t s_lo[8], s_hi[8];
    if (threadIdx.x < 8) {
        s_lo[threadIdx.x] = 0xFFFFFFFFu;
        s_hi[threadIdx.x] = 0u;
    }
    lrb_barrier();
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
        for (uint32_t d = 0; d < nd; ++d) {
            const uint32_t u = hdb_ord_bits(Xp[(uint64_t)i * dp + d]);
            atomicMin(&s_lo[d], u);
            atomicMax(&s_hi[d], u);
        }
    lrb_barrier();
    if (threadIdx.x < nd) {
        atomicMin(&mm[threadIdx.x], s_lo[threadIdx.x]);
        atomicMax(&mm[64 + threadIdx.x], s_hi[threadIdx.x]);
    }
}

__global__ __launch_bounds__(256) void hdb_morton_kernel(const float *__restrict__ Xp, uint32_t n, uint32_t dp, uint32_t nd,
                                                         const uint32_t *__restrict__ mm, uint64_t *__restrict__ keys,
                                                         uint32_t *__restrict__ vals)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t bits = 64u / nd < 16u ? 64u / nd : 16u; // 8 bits per coordinate at 8 dimensions
    uint32_t q[8];
    for (uint32_t d = 0; d < nd; ++d) {
        const float lo = hdb_ord_float(mm[d]), hi = hdb_ord_float(mm[64 + d]);
        const float span = hi - lo;
        float t = span > 0.0f ? (Xp[(uint64_t)i * dp + d] - lo) / span : 0.0f;
        t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
        uint32_t v = (uint32_t)(t * (float)(1u << bits));
        q[d] = v >= (1u << bits) ? (1u << bits) - 1u : v;
    }
    uint64_t key = 0;
    for (int b = (int)bits - 1; b >= 0; --b)
        for (uint32_t d = 0; d < nd; ++d) key = (key << 1) | (uint64_t)((q[d] >> b) & 1u);
    keys[i] = key;
    vals[i] = i;
}

__global__ __launch_bounds__(256) void hdb_gather_rows_kernel(const float *__restrict__ Xp, const uint32_t *__restrict__ ord,
                                                              uint64_t n, uint32_t dp, float *__restrict__ Xs)
{
    const uint64_t total = n * dp;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256) {
        const uint64_t r = i / dp;
        Xs[i] = Xp[(uint64_t)ord[r] * dp + (i - r * dp)];
    }
}

// box[b][0][d] = min, box[b][1][d] = max over rows [b * rows_per, (b + 1) * rows_per) of Xs
__global__ __launch_bounds__(256) void hdb_box_kernel(const float *__restrict__ Xs, uint32_t n, uint32_t dp, uint32_t rows_per,
                                                      uint32_t n_boxes, float *__restrict__ box)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x; // (box, dim)
    if (i >= n_boxes * dp) return;
    const uint32_t b = i / dp, d = i - b * dp;
    const uint32_t r0 = b * rows_per, r1 = r0 + rows_per < n ? r0 + rows_per : n;
    float lo = INFINITY, hi = -INFINITY;
    for (uint32_t r = r0; r < r1; ++r) {
        const float x = Xs[(uint64_t)r * dp + d];
        lo = fminf(lo, x);
        hi = fmaxf(hi, x);
    }
    box[((uint64_t)b * 2) * dp + d] = lo;
    box[((uint64_t)b * 2 + 1) * dp + d] = hi;
}

// WINDOW: the select over tiles [t0, t1) around the group's own; writes U2[group] = the largest k-th key of its
//         queries (uint bits of a squared distance), +inf when the window holds fewer than k rows.
// !WINDOW: the select over all tiles not farther than U2[group]; writes core[ord[q]].
template <int DP, bool WINDOW>
__global__ __launch_bounds__(256) void hdb_core_sel_kernel(const float *__restrict__ Xs, uint32_t n, uint32_t k,
                                                           const float *__restrict__ gbox, const float *__restrict__ tbox,
                                                           uint32_t window_tiles, uint32_t *__restrict__ U2,
                                                           const uint32_t *__restrict__ ord, float *__restrict__ core)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *hist = smem;                                         // [256 digits][64 lanes]
    float *tile = reinterpret_cast<float *>(smem + 256 * HDB_Q);   // [256 rows][DP]
    __shared__ uint32_t s_prefix[HDB_Q], s_krem[HDB_Q];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t g = blockIdx.x, q = g * HDB_Q + lane;
    float xq[DP];
    hdb_load_query<DP>(Xs, q < n ? q : n - 1, xq);
    const uint32_t n_tiles = (n + HDB_TILE - 1) / HDB_TILE;
    uint32_t t0 = 0, t1 = n_tiles;
    float thr = INFINITY;
    float glo[DP], ghi[DP];
    if (WINDOW) {
        const uint32_t own = (g * HDB_Q) / HDB_TILE, span = 2 * window_tiles + 1;
        t0 = own > window_tiles ? own - window_tiles : 0;
        t1 = t0 + span < n_tiles ? t0 + span : n_tiles;
        t0 = t1 > span ? t1 - span : 0;
        const uint64_t rows = (uint64_t)(t1 - t0) * HDB_TILE;
        if ((t1 == n_tiles ? (uint64_t)n - (uint64_t)t0 * HDB_TILE : rows) < k) { // uniform: no bound from this window
            if (tid == 0) U2[g] = 0x7F800000u;
            return;
        }
    } else {
        const uint32_t u = U2[g];
        thr = u >= 0x7F800000u ? INFINITY : __uint_as_float(u) * 1.00001f;
#pragma unroll
        for (int d = 0; d < DP; ++d) {
            glo[d] = gbox[((uint64_t)g * 2) * DP + d];
            ghi[d] = gbox[((uint64_t)g * 2 + 1) * DP + d];
        }
    }
    uint32_t prefix = 0, krem = k;
#pragma unroll 1
    for (int pass = 0; pass < 4; ++pass) {
        const uint32_t width = pass == 3 ? 7u : 8u;
        const uint32_t shift = pass == 0 ? 23u : pass == 1 ? 15u : pass == 2 ? 7u : 0u;
        const uint32_t hs = shift + width;
        const uint32_t wmask = (1u << width) - 1u;
        for (uint32_t i = tid; i < 256 * HDB_Q; i += 256) hist[i] = 0;
#pragma unroll 1
        for (uint32_t t = t0; t < t1; ++t) {
            if (!WINDOW) {
                // squared distance between the two boxes: a lower bound for every pair (uniform over the workgroup)
                float lb2 = 0.0f;
#pragma unroll
                for (int d = 0; d < DP; ++d) {
                    const float a = tbox[((uint64_t)t * 2) * DP + d] - ghi[d], b = glo[d] - tbox[((uint64_t)t * 2 + 1) * DP + d];
                    const float gap = fmaxf(fmaxf(a, b), 0.0f);
                    lb2 = fmaf(gap, gap, lb2);
                }
                if (lb2 > thr) continue;
            }
            const uint32_t tile0 = t * HDB_TILE;
            lrb_barrier();
            hdb_stage_tile<DP>(Xs, n, tile0, tile, tid);
            lrb_barrier();
            const uint32_t c0 = wave * 64u;
            const uint32_t left = n - tile0;
            const uint32_t cend = left < c0 + 64u ? (left > c0 ? left : c0) : c0 + 64u;
#pragma unroll 4
            for (uint32_t c = c0; c < cend; ++c) {
                const uint32_t key = __float_as_uint(hdb_dist2<DP>(xq, tile + c * DP));
                if ((hs >= 31u ? 0u : key >> hs) == prefix)
                    atomicAdd(&hist[((key >> shift) & wmask) * HDB_Q + lane], 1u);
            }
        }
        lrb_barrier();
        if (wave == 0) {
            uint32_t cum = 0, digit = wmask;
            for (uint32_t b = 0; b <= wmask; ++b) {
                const uint32_t cnt = hist[b * HDB_Q + lane];
                if (cum + cnt >= krem) {
                    digit = b;
                    break;
                }
                cum += cnt;
            }
            s_prefix[lane] = (prefix << width) | digit;
            s_krem[lane] = krem - cum;
        }
        lrb_barrier();
        prefix = s_prefix[lane];
        krem = s_krem[lane];
        lrb_barrier();
    }
    if (WINDOW) {
        // largest k-th key of the group's real queries (non-negative floats order like their bits)
        uint32_t m = q < n ? prefix : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t other = (uint32_t)__shfl_xor((int)m, o, 64);
            m = other > m ? other : m;
        }
        if (tid == 0) U2[g] = m;
    } else if (wave == 0 && q < n) {
        core[ord[q]] = sqrtf(__uint_as_float(prefix));
    }
}

template <int DP>
__global__ __launch_bounds__(256) void hdb_nearest_kernel(const float *__restrict__ Xp,
                                                          const float *__restrict__ core,
                                                          const uint32_t *__restrict__ comp, uint32_t n,
                                                          float *__restrict__ best_w,
                                                          uint32_t *__restrict__ best_j)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    float *tile = reinterpret_cast<float *>(smem);          // [256][DP]
    float *t_core2 = tile + HDB_TILE * DP;                  // [256]
    uint32_t *t_comp = reinterpret_cast<uint32_t *>(t_core2 + HDB_TILE);
    __shared__ float s_w[4][HDB_Q];
    __shared__ uint32_t s_j[4][HDB_Q];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t q = blockIdx.x * HDB_Q + lane;
    const uint32_t qq = q < n ? q : n - 1;
    float xq[DP];
    hdb_load_query<DP>(Xp, qq, xq);
    const float cq = core[qq] * core[qq];
    const uint32_t compq = comp[qq];
    float bw = INFINITY;
    uint32_t bj = 0xFFFFFFFFu;
#pragma unroll 1
    for (uint32_t tile0 = 0; tile0 < n; tile0 += HDB_TILE) {
        lrb_barrier();
        hdb_stage_tile<DP>(Xp, n, tile0, tile, tid);
        if (tile0 + tid < n) {
            const float cj = core[tile0 + tid];
            t_core2[tid] = cj * cj;
            t_comp[tid] = comp[tile0 + tid];
        }
        lrb_barrier();
        const uint32_t c0 = wave * 64u;
        const uint32_t left = n - tile0;
        const uint32_t cend = left < c0 + 64u ? (left > c0 ? left : c0) : c0 + 64u;
#pragma unroll 4
        for (uint32_t c = c0; c < cend; ++c) {
            const float d2 = hdb_dist2<DP>(xq, tile + c * DP);
            const float mr = fmaxf(fmaxf(d2, cq), t_core2[c]);
            // candidates come in ascending index: strict '<' keeps the first of equal weights
            if (t_comp[c] != compq && mr < bw) {
                bw = mr;
                bj = tile0 + c;
            }
        }
    }
    s_w[wave][lane] = bw;
    s_j[wave][lane] = bj;
    lrb_barrier();
    if (wave == 0 && q < n) {
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float ow = s_w[w][lane];
            const uint32_t oj = s_j[w][lane];
            if (ow < bw || (ow == bw && oj < bj)) {
                bw = ow;
                bj = oj;
            }
        }
        best_w[q] = bw; // squared mutual reachability
        best_j[q] = bj;
    }
}

// ---- the Boruvka step on the same spatial order ---------------------------------------------------
__global__ __launch_bounds__(256) void hdb_gather_u32_kernel(const uint32_t *__restrict__ src, const uint32_t *__restrict__ ord,
                                                             uint32_t n, uint32_t *__restrict__ dst)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[ord[i]];
}

__global__ __launch_bounds__(256) void hdb_gather_sq_kernel(const float *__restrict__ core, const uint32_t *__restrict__ ord,
                                                            uint32_t n, float *__restrict__ dst)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const float cj = core[ord[i]];
        dst[i] = cj * cj;
    }
}

// lo[b], hi[b] = smallest / largest component id among rows [b * rows_per, (b + 1) * rows_per)
__global__ __launch_bounds__(256) void hdb_comp_range_kernel(const uint32_t *__restrict__ comp_s, uint32_t n, uint32_t rows_per,
                                                             uint32_t n_boxes, uint32_t *__restrict__ lo, uint32_t *__restrict__ hi)
{
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= n_boxes) return;
    const uint32_t r0 = b * rows_per, r1 = r0 + rows_per < n ? r0 + rows_per : n;
    uint32_t a = 0xFFFFFFFFu, z = 0u;
    for (uint32_t r = r0; r < r1; ++r) {
        const uint32_t cc = comp_s[r];
        a = cc < a ? cc : a;
        z = cc > z ? cc : z;
    }
    lo[b] = a;
    hi[b] = z;
}

// One Boruvka step for the 64 (Morton-ordered) queries of a group: the same candidates win as in
// hdb_nearest_kernel -- lightest mutual-reachability edge into another component, equal weights to the lower
// ORIGINAL index -- but tiles are skipped when (a) every row of the tile and every query belong to one and the same
// component, or (b) the tile's box is farther from the group's box than the worst of the queries' current best
// edges (d_mreach >= d).  A first sweep over the tiles around the group's own gives that bound something to work with.
template <int DP>
__global__ __launch_bounds__(256) void hdb_nearest_pruned_kernel(const float *__restrict__ Xs, const float *__restrict__ core2_s,
                                                                 const uint32_t *__restrict__ comp_s, const uint32_t *__restrict__ ord,
                                                                 uint32_t n, const float *__restrict__ gbox, const float *__restrict__ tbox,
                                                                 const uint32_t *__restrict__ gc_lo, const uint32_t *__restrict__ gc_hi,
                                                                 const uint32_t *__restrict__ tc_lo, const uint32_t *__restrict__ tc_hi,
                                                                 uint32_t window_tiles, float *__restrict__ best_w,
                                                                 uint32_t *__restrict__ best_j)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    float *tile = reinterpret_cast<float *>(smem);          // [256][DP]
    float *t_core2 = tile + HDB_TILE * DP;                  // [256]
    uint32_t *t_comp = reinterpret_cast<uint32_t *>(t_core2 + HDB_TILE);
    uint32_t *t_id = t_comp + HDB_TILE;
    __shared__ float s_w[4][HDB_Q];
    __shared__ uint32_t s_j[4][HDB_Q];
    __shared__ float s_bound;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t g = blockIdx.x, q = g * HDB_Q + lane;
    const uint32_t qq = q < n ? q : n - 1;
    float xq[DP];
    hdb_load_query<DP>(Xs, qq, xq);
    const float cq = core2_s[qq];
    const uint32_t compq = comp_s[qq];
    float glo[DP], ghi[DP];
#pragma unroll
    for (int d = 0; d < DP; ++d) {
        glo[d] = gbox[((uint64_t)g * 2) * DP + d];
        ghi[d] = gbox[((uint64_t)g * 2 + 1) * DP + d];
    }
    const uint32_t gcl = gc_lo[g], gch = gc_hi[g];
    const uint32_t n_tiles = (n + HDB_TILE - 1) / HDB_TILE;
    const uint32_t own = (g * HDB_Q) / HDB_TILE, span = 2 * window_tiles + 1;
    uint32_t w0 = own > window_tiles ? own - window_tiles : 0;
    const uint32_t w1 = w0 + span < n_tiles ? w0 + span : n_tiles;
    w0 = w1 > span ? w1 - span : 0;
    float bw = INFINITY, thr = INFINITY;
    uint32_t bj = 0xFFFFFFFFu, since = 0;
    auto refresh_bound = [&]() { // uniform: every thread calls it at the same points
        s_w[wave][lane] = bw;
        lrb_barrier();
        if (wave == 0) {
            float m = fminf(fminf(s_w[0][lane], s_w[1][lane]), fminf(s_w[2][lane], s_w[3][lane]));
            m = q < n ? m : 0.0f;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
            if (lane == 0) s_bound = m;
        }
        lrb_barrier();
        const float b = s_bound;
        thr = b < INFINITY ? b * 1.00001f : INFINITY;
    };
#pragma unroll 1
    for (int phase = 0; phase < 2; ++phase) {
        const uint32_t t_begin = phase == 0 ? w0 : 0, t_end = phase == 0 ? w1 : n_tiles;
#pragma unroll 1
        for (uint32_t t = t_begin; t < t_end; ++t) {
            if (phase == 1 && t >= w0 && t < w1) continue; // swept in the first phase
            if (gcl == gch && tc_lo[t] == gcl && tc_hi[t] == gcl) continue; // one component on both sides: no edge here
            if (phase == 1) {
                float lb2 = 0.0f;
#pragma unroll
                for (int d = 0; d < DP; ++d) {
                    const float a = tbox[((uint64_t)t * 2) * DP + d] - ghi[d], b = glo[d] - tbox[((uint64_t)t * 2 + 1) * DP + d];
                    const float gap = fmaxf(fmaxf(a, b), 0.0f);
                    lb2 = fmaf(gap, gap, lb2);
                }
                if (lb2 > thr) continue;
            }
            const uint32_t tile0 = t * HDB_TILE;
            lrb_barrier();
            hdb_stage_tile<DP>(Xs, n, tile0, tile, tid);
            if (tile0 + tid < n) {
                t_core2[tid] = core2_s[tile0 + tid];
                t_comp[tid] = comp_s[tile0 + tid];
                t_id[tid] = ord[tile0 + tid];
            }
            lrb_barrier();
            const uint32_t c0 = wave * 64u;
            const uint32_t left = n - tile0;
            const uint32_t cend = left < c0 + 64u ? (left > c0 ? left : c0) : c0 + 64u;
#pragma unroll 4
            for (uint32_t c = c0; c < cend; ++c) {
                const float d2 = hdb_dist2<DP>(xq, tile + c * DP);
                const float mr = fmaxf(fmaxf(d2, cq), t_core2[c]);
                const uint32_t id = t_id[c];
                if (t_comp[c] != compq && (mr < bw || (mr == bw && id < bj))) {
                    bw = mr;
                    bj = id;
                }
            }
            if (phase == 1 && ++since == 16) {
                since = 0;
                refresh_bound();
            }
        }
        if (phase == 0) refresh_bound();
    }
    lrb_barrier();
    s_w[wave][lane] = bw;
    s_j[wave][lane] = bj;
    lrb_barrier();
    if (wave == 0 && q < n) {
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float ow = s_w[w][lane];
            const uint32_t oj = s_j[w][lane];
            if (ow < bw || (ow == bw && oj < bj)) {
                bw = ow;
                bj = oj;
            }
        }
        const uint32_t Q = ord[q];
        best_w[Q] = bw; // squared mutual reachability
        best_j[Q] = bj;
    }
}

static uint32_t hdb_pad_dims(int dims) { return dims <= 4 ? 4u : dims <= 8 ? 8u : dims <= 16 ? 16u : dims <= 32 ? 32u : 64u; }

static int hdb_padded(lrb_ctx *c, const float *d_X, uint64_t n, int dims, uint32_t *dp_out, float **d_Xp)
{
    const uint32_t dp = hdb_pad_dims(dims);
    void *p;
    int rc = lrb_ws_get(c, 12, n * dp * sizeof(float), &p);
    if (rc != LRB_OK) return rc;
    uint64_t blocks = (n * dp + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(hdb_pad_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream, d_X, n, (uint32_t)dims, dp,
                       (float *)p);
    HIP_TRY(hipGetLastError());
    *dp_out = dp;
    *d_Xp = (float *)p;
    return LRB_OK;
}

template <int DP> static int hdb_launch_core(lrb_ctx *c, const float *Xp, uint32_t n, uint32_t k, float *d_core)
{
    const size_t smem = (size_t)256 * HDB_Q * 4 + (size_t)HDB_TILE * DP * 4;
    HIP_TRY(hipFuncSetAttribute((const void *)hdb_core_kernel<DP>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem));
    hipLaunchKernelGGL(hdb_core_kernel<DP>, dim3((n + HDB_Q - 1) / HDB_Q), dim3(256), smem, c->stream, Xp, n, k,
                       d_core);
    HIP_TRY(hipGetLastError());
    return LRB_OK;
}

// below this many points the brute-force select is as fast as the set-up of the pruned one
#define HDB_PRUNE_MIN 60000u

namespace {
struct hdb_scratch { // device allocations of one call
    std::vector<void *> p;
    ~hdb_scratch()
    {
        for (void *x : p) (void)hipFree(x);
    }
    template <typename T> int get(T **out, size_t count)
    {
        void *x = nullptr;
        HIP_TRY(hipMalloc(&x, count * sizeof(T) + 256));
        p.push_back(x);
        *out = (T *)x;
        return LRB_OK;
    }
};
} // namespace

// Morton order of the rows + the boxes of its 64-row groups and 256-row tiles (device memory owned by sc)
struct hdb_spatial {
    float *Xs = nullptr, *gbox = nullptr, *tbox = nullptr;
    uint32_t *ord = nullptr;
    uint32_t n_groups = 0, n_tiles = 0;
};

template <int DP> static int hdb_spatial_setup(lrb_ctx *c, const float *Xp, uint32_t n, int dims, hdb_scratch &sc, hdb_spatial &sp)
{
    const uint32_t nd = dims < 8 ? (uint32_t)dims : 8u;
    sp.n_groups = (n + HDB_Q - 1) / HDB_Q;
    sp.n_tiles = (n + HDB_TILE - 1) / HDB_TILE;
    uint32_t *mm, *vals;
    uint64_t *keys, *keys2;
    int rc;
    if ((rc = sc.get(&mm, 128)) != LRB_OK || (rc = sc.get(&keys, n)) != LRB_OK || (rc = sc.get(&keys2, n)) != LRB_OK ||
        (rc = sc.get(&vals, n)) != LRB_OK || (rc = sc.get(&sp.ord, n)) != LRB_OK || (rc = sc.get(&sp.Xs, (size_t)n * DP)) != LRB_OK ||
        (rc = sc.get(&sp.gbox, (size_t)sp.n_groups * 2 * DP)) != LRB_OK || (rc = sc.get(&sp.tbox, (size_t)sp.n_tiles * 2 * DP)) != LRB_OK)
        return rc;
    hipStream_t st = c->stream;
    HIP_TRY(hipMemsetAsync(mm, 0xFF, 64 * 4, st));
    HIP_TRY(hipMemsetAsync(mm + 64, 0, 64 * 4, st));
    const unsigned nb = (n + 255) / 256;
    hipLaunchKernelGGL(hdb_minmax_kernel, dim3(nb < 1024 ? nb : 1024), dim3(256), 0, st, Xp, n, (uint32_t)DP, nd, mm);
    hipLaunchKernelGGL(hdb_morton_kernel, dim3(nb), dim3(256), 0, st, Xp, n, (uint32_t)DP, nd, mm, keys, vals);
    {
        size_t tmp_bytes = 0;
        HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys, keys2, vals, sp.ord, (size_t)n, 0, 64, st));
        char *tmp;
        if ((rc = sc.get(&tmp, tmp_bytes)) != LRB_OK) return rc;
        HIP_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, keys2, vals, sp.ord, (size_t)n, 0, 64, st));
    }
    hipLaunchKernelGGL(hdb_gather_rows_kernel, dim3(nb * DP < 65535u ? nb * DP : 65535u), dim3(256), 0, st, Xp, sp.ord, (uint64_t)n,
                       (uint32_t)DP, sp.Xs);
    hipLaunchKernelGGL(hdb_box_kernel, dim3((sp.n_groups * DP + 255) / 256), dim3(256), 0, st, sp.Xs, n, (uint32_t)DP, (uint32_t)HDB_Q,
                       sp.n_groups, sp.gbox);
    hipLaunchKernelGGL(hdb_box_kernel, dim3((sp.n_tiles * DP + 255) / 256), dim3(256), 0, st, sp.Xs, n, (uint32_t)DP, (uint32_t)HDB_TILE,
                       sp.n_tiles, sp.tbox);
    HIP_TRY(hipGetLastError());
    return LRB_OK;
}

static uint32_t hdb_window()
{
    if (const char *e = getenv("LRB_HDB_WINDOW")) return (uint32_t)atoi(e);
    return 8;
}

template <int DP> static int hdb_launch_core_pruned(lrb_ctx *c, const float *Xp, uint32_t n, int dims, uint32_t k, float *d_core)
{
    hdb_scratch sc;
    hdb_spatial sp;
    int rc = hdb_spatial_setup<DP>(c, Xp, n, dims, sc, sp);
    if (rc != LRB_OK) return rc;
    uint32_t *U2;
    if ((rc = sc.get(&U2, sp.n_groups)) != LRB_OK) return rc;
    hipStream_t st = c->stream;
    const size_t smem = (size_t)256 * HDB_Q * 4 + (size_t)HDB_TILE * DP * 4;
    HIP_TRY(hipFuncSetAttribute((const void *)hdb_core_sel_kernel<DP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    HIP_TRY(hipFuncSetAttribute((const void *)hdb_core_sel_kernel<DP, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const uint32_t window = hdb_window();
    hipLaunchKernelGGL((hdb_core_sel_kernel<DP, true>), dim3(sp.n_groups), dim3(256), smem, st, sp.Xs, n, k, sp.gbox, sp.tbox, window, U2,
                       sp.ord, d_core);
    hipLaunchKernelGGL((hdb_core_sel_kernel<DP, false>), dim3(sp.n_groups), dim3(256), smem, st, sp.Xs, n, k, sp.gbox, sp.tbox, window, U2,
                       sp.ord, d_core);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st)); // the scratch is freed on return
    return LRB_OK;
}

// the per-round part of the pruned Boruvka step; comp_s / ranges are scratch of the caller
struct hdb_round_bufs {
    float *core2_s = nullptr;
    uint32_t *comp_s = nullptr, *gc_lo = nullptr, *gc_hi = nullptr, *tc_lo = nullptr, *tc_hi = nullptr;
};

template <int DP>
static int hdb_launch_nearest_pruned(lrb_ctx *c, const hdb_spatial &sp, const hdb_round_bufs &rb, const uint32_t *d_comp, uint32_t n,
                                     float *d_bw, uint32_t *d_bj)
{
    hipStream_t st = c->stream;
    const unsigned nb = (n + 255) / 256;
    hipLaunchKernelGGL(hdb_gather_u32_kernel, dim3(nb), dim3(256), 0, st, d_comp, sp.ord, n, rb.comp_s);
    hipLaunchKernelGGL(hdb_comp_range_kernel, dim3((sp.n_groups + 255) / 256), dim3(256), 0, st, rb.comp_s, n, (uint32_t)HDB_Q, sp.n_groups,
                       rb.gc_lo, rb.gc_hi);
    hipLaunchKernelGGL(hdb_comp_range_kernel, dim3((sp.n_tiles + 255) / 256), dim3(256), 0, st, rb.comp_s, n, (uint32_t)HDB_TILE, sp.n_tiles,
                       rb.tc_lo, rb.tc_hi);
    const size_t smem = (size_t)HDB_TILE * DP * 4 + HDB_TILE * 12;
    HIP_TRY(hipFuncSetAttribute((const void *)hdb_nearest_pruned_kernel<DP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(hdb_nearest_pruned_kernel<DP>, dim3(sp.n_groups), dim3(256), smem, st, sp.Xs, rb.core2_s, rb.comp_s, sp.ord, n, sp.gbox,
                       sp.tbox, rb.gc_lo, rb.gc_hi, rb.tc_lo, rb.tc_hi, hdb_window(), d_bw, d_bj);
    HIP_TRY(hipGetLastError());
    return LRB_OK;
}

template <int DP>
static int hdb_launch_nearest(lrb_ctx *c, const float *Xp, const float *d_core, const uint32_t *d_comp, uint32_t n,
                              float *d_bw, uint32_t *d_bj)
{
    const size_t smem = (size_t)HDB_TILE * DP * 4 + HDB_TILE * 8;
    HIP_TRY(hipFuncSetAttribute((const void *)hdb_nearest_kernel<DP>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem));
    hipLaunchKernelGGL(hdb_nearest_kernel<DP>, dim3((n + HDB_Q - 1) / HDB_Q), dim3(256), smem, c->stream, Xp, d_core,
                       d_comp, n, d_bw, d_bj);
    HIP_TRY(hipGetLastError());
    return LRB_OK;
}

#define HDB_DISPATCH(dp, fn, ...)                                                   \
    ((dp) == 4 ? fn<4>(__VA_ARGS__)                                                 \
               : (dp) == 8 ? fn<8>(__VA_ARGS__)                                     \
                           : (dp) == 16 ? fn<16>(__VA_ARGS__)                       \
                                        : (dp) == 32 ? fn<32>(__VA_ARGS__) : fn<64>(__VA_ARGS__))

extern "C" int lrb_hdb_core_dist_dev(lrb_ctx *c, const float *d_X, uint64_t n, int dims, uint32_t k, float *d_core)
{
    ARG_TRY(c != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    ARG_TRY(dims >= 1 && dims <= 64);
    if (n == 0) return LRB_OK;
    ARG_TRY(d_X && d_core);
    ARG_TRY(n < 0x7FFFFFFFull);
    ARG_TRY(k >= 1 && k <= n);
    uint32_t dp;
    float *Xp;
    int rc = hdb_padded(c, d_X, n, dims, &dp, &Xp);
    if (rc != LRB_OK) return rc;
    const bool brute = n < HDB_PRUNE_MIN || (getenv("LRB_HDB_BRUTE") && atoi(getenv("LRB_HDB_BRUTE")));
    if (brute) return HDB_DISPATCH(dp, hdb_launch_core, c, Xp, (uint32_t)n, k, d_core);
    return HDB_DISPATCH(dp, hdb_launch_core_pruned, c, Xp, (uint32_t)n, dims, k, d_core);
}

namespace {
struct uf_t {
    std::vector<uint32_t> p;
    explicit uf_t(uint32_t n) : p(n)
    {
        for (uint32_t i = 0; i < n; ++i) p[i] = i;
    }
    uint32_t find(uint32_t x)
    {
        while (p[x] != x) {
            p[x] = p[p[x]];
            x = p[x];
        }
        return x;
    }
};
struct cand_t {
    float w;
    uint32_t lo, hi;
    bool valid;
};
inline bool cand_less(float w, uint32_t lo, uint32_t hi, const cand_t &b)
{
    if (!b.valid) return true;
    if (w != b.w) return w < b.w;
    if (lo != b.lo) return lo < b.lo;
    return hi < b.hi;
}
} // namespace

extern "C" int lrb_hdb_mst_dev(lrb_ctx *c, const float *d_X, uint64_t n64, int dims, const float *d_core,
                               uint32_t *h_u, uint32_t *h_v, float *h_w, uint32_t *rounds_out)
{
    ARG_TRY(c != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    ARG_TRY(dims >= 1 && dims <= 64);
    if (rounds_out) *rounds_out = 0;
    if (n64 <= 1) return LRB_OK;
    ARG_TRY(d_X && d_core && h_u && h_v && h_w);
    ARG_TRY(n64 < 0x7FFFFFFFull);
    const uint32_t n = (uint32_t)n64;
    uint32_t dp;
    float *Xp;
    int rc = hdb_padded(c, d_X, n, dims, &dp, &Xp);
    if (rc != LRB_OK) return rc;
    void *p_comp, *p_bw, *p_bj;
    if ((rc = lrb_ws_get(c, 13, (uint64_t)n * 4, &p_comp)) != LRB_OK) return rc;
    if ((rc = lrb_ws_get(c, 14, (uint64_t)n * 4, &p_bw)) != LRB_OK) return rc;
    if ((rc = lrb_ws_get(c, 15, (uint64_t)n * 4, &p_bj)) != LRB_OK) return rc;
    uint32_t *d_comp = (uint32_t *)p_comp;
    float *d_bw = (float *)p_bw;
    uint32_t *d_bj = (uint32_t *)p_bj;

    // large inputs: the Boruvka steps run on the Morton order with box / component pruning (same edges)
    const bool brute = n < HDB_PRUNE_MIN || (getenv("LRB_HDB_BRUTE") && atoi(getenv("LRB_HDB_BRUTE")));
    hdb_scratch sc;
    hdb_spatial sp;
    hdb_round_bufs rb;
    if (!brute) {
        rc = HDB_DISPATCH(dp, hdb_spatial_setup, c, Xp, n, dims, sc, sp);
        if (rc != LRB_OK) return rc;
        if ((rc = sc.get(&rb.core2_s, n)) != LRB_OK || (rc = sc.get(&rb.comp_s, n)) != LRB_OK || (rc = sc.get(&rb.gc_lo, sp.n_groups)) != LRB_OK ||
            (rc = sc.get(&rb.gc_hi, sp.n_groups)) != LRB_OK || (rc = sc.get(&rb.tc_lo, sp.n_tiles)) != LRB_OK ||
            (rc = sc.get(&rb.tc_hi, sp.n_tiles)) != LRB_OK)
            return rc;
        hipLaunchKernelGGL(hdb_gather_sq_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_core, sp.ord, n, rb.core2_s);
        HIP_TRY(hipGetLastError());
    }
    uf_t uf(n);
    std::vector<uint32_t> comp(n), bj(n);
    std::vector<float> bw(n);
    std::vector<cand_t> cmin(n);
    for (uint32_t i = 0; i < n; ++i) comp[i] = i;
    uint32_t n_edges = 0, rounds = 0;
    while (n_edges + 1 < n) {
        HIP_TRY(hipMemcpyAsync(d_comp, comp.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
        if (brute)
            rc = HDB_DISPATCH(dp, hdb_launch_nearest, c, Xp, d_core, d_comp, n, d_bw, d_bj);
        else
            rc = HDB_DISPATCH(dp, hdb_launch_nearest_pruned, c, sp, rb, d_comp, n, d_bw, d_bj);
        if (rc != LRB_OK) return rc;
        HIP_TRY(hipMemcpyAsync(bw.data(), d_bw, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(bj.data(), d_bj, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        ++rounds;
        // lightest outgoing edge of every component under (w, lo, hi)
        for (uint32_t i = 0; i < n; ++i) cmin[i].valid = false;
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t j = bj[i];
            if (j == 0xFFFFFFFFu) continue;
            const uint32_t lo = i < j ? i : j, hi = i < j ? j : i;
            cand_t &m = cmin[comp[i]];
            if (cand_less(bw[i], lo, hi, m)) m = {bw[i], lo, hi, true};
        }
        const uint32_t before = n_edges;
        for (uint32_t r = 0; r < n; ++r) {
            if (!cmin[r].valid) continue;
            const uint32_t a = uf.find(cmin[r].lo), b = uf.find(cmin[r].hi);
            if (a == b) continue; // the same edge chosen from both sides
            uf.p[a < b ? b : a] = a < b ? a : b;
            h_u[n_edges] = cmin[r].lo;
            h_v[n_edges] = cmin[r].hi;
            h_w[n_edges] = sqrtf(cmin[r].w);
            ++n_edges;
        }
        if (n_edges == before) {
            lrb_set_error("Boruvka round added no edge (non-finite coordinates?)%s%s", "", "");
            return LRB_ERR_ARG;
        }
        for (uint32_t i = 0; i < n; ++i) comp[i] = uf.find(i);
    }
    if (rounds_out) *rounds_out = rounds;
    return LRB_OK;
}

// Which neighbour the core distance is taken to: the k-th with the point itself counted
// (sklearn.cluster.HDBSCAN; the hdbscan package's Prim's paths: tree.query(X, k = min_samples)[:, -1]) or the k-th
// OTHER point (the package's Boruvka paths -- what algorithm='best' takes for euclidean latents of <= 60 dimensions,
// i.e. the reference's call: tree.query(X, k = min_samples + 1)[:, min_samples]).  Default: the second; LRB_HDB_CORE=self
// selects the first.
static int hdb_default_excludes_self()
{
    const char *e = getenv("LRB_HDB_CORE");
    return !(e && (e[0] == 's' || e[0] == 'S'));
}

extern "C" int lrb_hdbscan_host_ex(lrb_ctx *c, const float *X, uint64_t n, int dims, uint32_t min_cluster_size,
                                   uint32_t min_samples, int core_excludes_self, int32_t *labels, uint32_t *n_clusters)
{
    ARG_TRY(c != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    ARG_TRY(dims >= 1 && dims <= 64);
    ARG_TRY(min_cluster_size >= 2);
    if (n_clusters) *n_clusters = 0;
    if (n == 0) return LRB_OK;
    ARG_TRY(X && labels);
    ARG_TRY(min_samples >= 1);
    if (core_excludes_self < 0) core_excludes_self = hdb_default_excludes_self();
    // the package: min_samples = min(n - 1, min_samples), at least 1 (hdbscan_.py, hdbscan()): fewer points than
    // min_samples is not an error there
    uint64_t ms = min_samples < n - 1 ? min_samples : n - 1;
    if (ms == 0) ms = 1;
    uint64_t k = core_excludes_self ? ms + 1 : ms;
    if (k > n) k = n;
    void *p_x, *p_core;
    int rc;
    if ((rc = lrb_ws_get(c, 0, n * dims * sizeof(float), &p_x)) != LRB_OK) return rc;
    if ((rc = lrb_ws_get(c, 1, n * sizeof(float), &p_core)) != LRB_OK) return rc;
    HIP_TRY(hipMemcpyAsync(p_x, X, n * dims * sizeof(float), hipMemcpyHostToDevice, c->stream));
    rc = lrb_hdb_core_dist_dev(c, (const float *)p_x, n, dims, (uint32_t)k, (float *)p_core);
    if (rc != LRB_OK) return rc;
    if (n == 1) {
        labels[0] = -1;
        HIP_TRY(hipStreamSynchronize(c->stream));
        return LRB_OK;
    }
    std::vector<uint32_t> u(n - 1), v(n - 1);
    std::vector<float> w(n - 1);
    rc = lrb_hdb_mst_dev(c, (const float *)p_x, n, dims, (const float *)p_core, u.data(), v.data(), w.data(), nullptr);
    if (rc != LRB_OK) return rc;
    return lrb_hdb_labels(n, u.data(), v.data(), w.data(), min_cluster_size, labels, n_clusters);
}

extern "C" int lrb_hdbscan_host(lrb_ctx *c, const float *X, uint64_t n, int dims, uint32_t min_cluster_size,
                                uint32_t min_samples, int32_t *labels, uint32_t *n_clusters)
{
    return lrb_hdbscan_host_ex(c, X, n, dims, min_cluster_size, min_samples, -1, labels, n_clusters);
}
