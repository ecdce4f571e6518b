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
            __syncthreads();
            hdb_stage_tile<DP>(Xp, n, tile0, tile, tid);
            __syncthreads();
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
        __syncthreads();
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
        __syncthreads();
        prefix = s_prefix[lane];
        krem = s_krem[lane];
        __syncthreads();
    }
    if (wave == 0 && q < n) core[q] = sqrtf(__uint_as_float(prefix));
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
        __syncthreads();
        hdb_stage_tile<DP>(Xp, n, tile0, tile, tid);
        if (tile0 + tid < n) {
            const float cj = core[tile0 + tid];
            t_core2[tid] = cj * cj;
            t_comp[tid] = comp[tile0 + tid];
        }
        __syncthreads();
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
    __syncthreads();
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
    ARG_TRY(dims >= 1 && dims <= 64);
    if (n == 0) return LRB_OK;
    ARG_TRY(d_X && d_core);
    ARG_TRY(n < 0x7FFFFFFFull);
    ARG_TRY(k >= 1 && k <= n);
    uint32_t dp;
    float *Xp;
    int rc = hdb_padded(c, d_X, n, dims, &dp, &Xp);
    if (rc != LRB_OK) return rc;
    return HDB_DISPATCH(dp, hdb_launch_core, c, Xp, (uint32_t)n, k, d_core);
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

    uf_t uf(n);
    std::vector<uint32_t> comp(n), bj(n);
    std::vector<float> bw(n);
    std::vector<cand_t> cmin(n);
    for (uint32_t i = 0; i < n; ++i) comp[i] = i;
    uint32_t n_edges = 0, rounds = 0;
    while (n_edges + 1 < n) {
        HIP_TRY(hipMemcpyAsync(d_comp, comp.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
        rc = HDB_DISPATCH(dp, hdb_launch_nearest, c, Xp, d_core, d_comp, n, d_bw, d_bj);
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

extern "C" int lrb_hdbscan_host(lrb_ctx *c, const float *X, uint64_t n, int dims, uint32_t min_cluster_size,
                                uint32_t min_samples, int32_t *labels, uint32_t *n_clusters)
{
    ARG_TRY(c != nullptr);
    ARG_TRY(dims >= 1 && dims <= 64);
    ARG_TRY(min_cluster_size >= 2);
    if (n_clusters) *n_clusters = 0;
    if (n == 0) return LRB_OK;
    ARG_TRY(X && labels);
    ARG_TRY(min_samples >= 1 && min_samples <= n);
    void *p_x, *p_core;
    int rc;
    if ((rc = lrb_ws_get(c, 0, n * dims * sizeof(float), &p_x)) != LRB_OK) return rc;
    if ((rc = lrb_ws_get(c, 1, n * sizeof(float), &p_core)) != LRB_OK) return rc;
    HIP_TRY(hipMemcpyAsync(p_x, X, n * dims * sizeof(float), hipMemcpyHostToDevice, c->stream));
    rc = lrb_hdb_core_dist_dev(c, (const float *)p_x, n, dims, min_samples, (float *)p_core);
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
