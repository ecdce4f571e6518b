// lrb_cluster.hip -- the kernels behind the clustering stage and their C ABI:
//   K4  seed_dist_kernel   0.5 - M @ M[seed]                                   (cluster_utils.py:45-49)
//       seed_hist_kernel   torch.histc(0.5 - M @ M[s], 60, 0, 0.3) for S seeds (cluster_utils.py:136-192)
//   K5  gauss_assign_kernel  left-over reads against the clusters' Gaussians   (cluster_utils.py:261-268,309-322)
#include "lrb_device.h"

namespace {
constexpr int WAVE = 64;
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

int grid_for_waves(const lrb_ctx *c, uint64_t n_waves_wanted, int waves_per_block, int blocks_per_cu)
{
    uint64_t blocks = (n_waves_wanted + waves_per_block - 1) / waves_per_block;
    const uint64_t cap = (uint64_t)c->n_cu * blocks_per_cu;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}
} // namespace

// ---------------------------------------------------------------------------
// K4: clustering distances.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void seed_dist_kernel(const float *__restrict__ M, uint64_t n,
                                                        int dims, uint64_t seed,
                                                        float *__restrict__ out)
{
    __shared__ float s[64];
    if (threadIdx.x < (uint32_t)dims) s[threadIdx.x] = M[seed * dims + threadIdx.x];
    lrb_barrier();
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const float *row = M + i * dims;
        float acc = 0.f;
        for (int k = 0; k < dims; ++k) acc = __builtin_fmaf(row[k], s[k], acc);
        out[i] = (i == seed) ? 0.f : 0.5f - acc;
    }
}

// torch.histc(x, 60, 0, 0.3) bin of one element (float32 arithmetic in torch's order:
// (x - lo) * nbins / (hi - lo), truncate, last edge inclusive); -1 = outside.
__device__ __forceinline__ int histc_bin(float x)
{
    const float lo = 0.0f, hi = 0.3f;
    if (!(x >= lo) || !(x <= hi)) return -1;
    int pos = (int)(((x - lo) * 60.0f) / (hi - lo));
    if (pos >= LRB_HIST_BINS) pos = LRB_HIST_BINS - 1;
    return pos;
}

// K4 (round 5): a SEED PER LANE.  A workgroup owns 256 seeds -- lane l of wave w holds the row of seed s0 + 64 w + l in
// registers -- and a chunk of the points; the points' rows are wave-uniform (scalar loads: the same address for all 64
// lanes), so a (point, seed) pair costs its dims FMAs, the bin arithmetic and ONE LDS operation: the increment of the lane's
// own histogram, laid out h[bin][lane] so that a wave's 64 increments fall in 64 different banks whatever bins they hit (the
// round-1 kernel walked seeds per point with four scattered LDS reads of the seed row + the atomic, seeds 60 words apart:
// 4-way bank aliasing; 0.44 ms at N = 432,333 / S = 1,000).
//   * the bin is torch.histc's: (int)(((x - 0) * 60) / 0.3f), the division done as q = y R, r = fma(-q, 0.3f, y), q' = fma(r, R, q)
//     with R = RN(1 / 0.3f) -- the same INTEGER PART as the correctly rounded quotient for every float y in [0, 18.1]
//     (scripts/k4_divcheck.c walks all 1.1e9 of them: the quotients differ for 3.7 M denormal y only, the bins never);
//   * in range <=> the bits of d, as unsigned, are at most those of 0.3f (d is never -0.0: 0.5 - acc rounds to +0, NaN and
//     negative values have larger bit patterns): the bits are CLAMPED to those of the next float after 0.3f, which the same
//     arithmetic sends to 60 -- a row of the histogram nobody reads -- while no d in the range reaches 60 (0.3f itself
//     gives 59: the checker walks every d too), so there is no compare, no select and no clamp of the bin;
//     (the increments made under the range test's exec mask instead -- no row 60, no clamp -- are SLOWER whatever share of the
//     pairs is in range: 0.284 against 0.190 ms in same-box pairs, profiles/r05_k4_ab.txt: the branches cost more than the
//     LDS operations they skip)
//   * a seed's own point counts as distance 0 (cluster_utils.py:48): the loop treats it like any other point and the lane
//     moves that one tally from where the arithmetic put it to bin 0 afterwards (it knows both: same FMA order).
// DIMS = the row length when it is 1..8 (registers), else MAXD = 16 / 32 / 64 registers with the row length at run time.
#define SEEDS_PER_WG 256
#define SEED_PB 8 // points a block of the main loop: the NEXT block's rows are asked for (scalar loads) before this one's pairs
template <int DIMS, int MAXD>
__global__ __launch_bounds__(256) void seed_hist_kernel(const float *__restrict__ M, uint64_t n, int dims_rt,
                                                        const int64_t *__restrict__ seeds, uint32_t n_seeds,
                                                        uint32_t chunk, uint32_t *__restrict__ part)
{
    constexpr int NR = DIMS > 0 ? DIMS : MAXD;
    const int dims = DIMS > 0 ? DIMS : dims_rt;
    // h[bin][lane], bin 60 = the tallies outside [0, 0.3] (nobody reads it: the increment needs no branch)
    __shared__ uint32_t h[(LRB_HIST_BINS + 1) * SEEDS_PER_WG];
    const uint32_t tid = threadIdx.x;
    const uint32_t s = blockIdx.y * SEEDS_PER_WG + tid;
    const bool live = s < n_seeds;
#pragma unroll
    for (int b = 0; b <= LRB_HIST_BINS; ++b) h[b * SEEDS_PER_WG + tid] = 0; // (the lane's own words: no barrier anywhere)
    const uint64_t sid = live ? (uint64_t)seeds[s] : 0;
    float sr[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) sr[k] = (live && k < dims) ? M[sid * dims + k] : 0.f;
    const uint64_t p0 = (uint64_t)blockIdx.x * chunk;
    const uint64_t p1 = p0 + chunk < n ? p0 + chunk : n;
    const float R = 1.0f / 0.3f; // RN(1 / 0.3f), folded at compile time
    const uint32_t top = __float_as_uint(0.3f);

    auto bin_of = [&](float d) -> uint32_t { // 60: outside [0, 0.3]
        const uint32_t u = __float_as_uint(d);
        const float dc = __uint_as_float(u < top + 1u ? u : top + 1u);
        const float y = dc * 60.0f;
        const float q = y * R;
        const float r = __builtin_fmaf(-q, 0.3f, y);
        return (uint32_t)(int)__builtin_fmaf(r, R, q);
    };
    auto tally = [&](float acc) { atomicAdd(&h[bin_of(0.5f - acc) * SEEDS_PER_WG + tid], 1u); }; // no return value: ds_add_u32

    uint64_t i = p0;
    if (DIMS > 0) {
        // rows are wave-uniform: the loads are scalar loads into SGPRs; a block's loads are issued one block ahead.  The
        // pairs of TWO points go through the arithmetic side by side (float2: v_pk_fma_f32 / v_pk_mul_f32, two lanes' worth of
        // work an instruction; every lane's own chain of FMAs is the same as before)
        typedef float f2 __attribute__((ext_vector_type(2)));
        static_assert(SEED_PB % 2 == 0, "points go in twos");
        float cur[SEED_PB][NR], nxt[SEED_PB][NR];
        if (i + SEED_PB <= p1) {
#pragma unroll
            for (int p = 0; p < SEED_PB; ++p)
#pragma unroll
                for (int k = 0; k < NR; ++k) cur[p][k] = M[(i + p) * DIMS + k];
        }
        for (; i + SEED_PB <= p1; i += SEED_PB) {
            // (the block after the last one: the same rows again -- a load nobody waits for, no branch in the body)
            const uint64_t j = i + 2 * SEED_PB <= p1 ? i + SEED_PB : i;
#pragma unroll
            for (int p = 0; p < SEED_PB; ++p)
#pragma unroll
                for (int k = 0; k < NR; ++k) nxt[p][k] = M[(j + p) * DIMS + k];
#pragma unroll
            for (int p = 0; p < SEED_PB; p += 2) {
                f2 acc = {0.f, 0.f};
#pragma unroll
                for (int k = 0; k < NR; ++k)
                    acc = __builtin_elementwise_fma((f2){cur[p][k], cur[p + 1][k]}, (f2){sr[k], sr[k]}, acc);
                const f2 d = (f2){0.5f, 0.5f} - acc;
                const uint32_t u0 = __float_as_uint(d.x), u1 = __float_as_uint(d.y);
                const f2 dc = {__uint_as_float(u0 < top + 1u ? u0 : top + 1u), __uint_as_float(u1 < top + 1u ? u1 : top + 1u)};
                const f2 y = dc * 60.0f;
                const f2 q = y * R;
                const f2 r = __builtin_elementwise_fma(-q, (f2){0.3f, 0.3f}, y);
                const f2 q2 = __builtin_elementwise_fma(r, (f2){R, R}, q);
                atomicAdd(&h[(uint32_t)(int)q2.x * SEEDS_PER_WG + tid], 1u);
                atomicAdd(&h[(uint32_t)(int)q2.y * SEEDS_PER_WG + tid], 1u);
            }
#pragma unroll
            for (int p = 0; p < SEED_PB; ++p)
#pragma unroll
                for (int k = 0; k < NR; ++k) cur[p][k] = nxt[p][k];
        }
    }
    for (; i < p1; ++i) { // the chunk's last points (and every point when the row length is a run-time value)
        const float *__restrict__ row = M + i * dims;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < NR; ++k)
            if (DIMS > 0 || k < dims) acc = __builtin_fmaf(row[k], sr[k], acc);
        tally(acc);
    }
    if (live && sid >= p0 && sid < p1) { // the seed's own point: whatever the loop made of it, it is distance 0
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < NR; ++k)
            if (DIMS > 0 || k < dims) acc = __builtin_fmaf(sr[k], sr[k], acc);
        h[bin_of(0.5f - acc) * SEEDS_PER_WG + tid] -= 1u;
        h[tid] += 1u;
    }
    // the workgroup's tallies leave as they lie, part[chunk][bin][seed]: coalesced plain stores, summed over the chunks by
    // seed_hist_sum_kernel (512 workgroups x 256 x 60 scattered global atomics took as long as the pairs)
    const uint32_t stride = gridDim.y * SEEDS_PER_WG;
    uint32_t *out = part + (uint64_t)blockIdx.x * LRB_HIST_BINS * stride + s;
#pragma unroll 4
    for (int b = 0; b < LRB_HIST_BINS; ++b) out[(uint64_t)b * stride] = h[b * SEEDS_PER_WG + tid];
}

// hist[s][b] = sum over the chunks of part[chunk][b][s]: a thread per (bin, seed), lanes along the seeds (coalesced reads)
__global__ __launch_bounds__(256) void seed_hist_sum_kernel(const uint32_t *__restrict__ part, uint32_t chunks, uint32_t stride,
                                                            uint32_t n_seeds, uint32_t *__restrict__ hist)
{
    const uint32_t s = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (s >= n_seeds) return;
    const uint32_t *p = part + (uint64_t)b * stride + s;
    const uint64_t step = (uint64_t)LRB_HIST_BINS * stride;
    uint32_t sum = 0;
    uint32_t c = 0;
    for (; c + 8 <= chunks; c += 8) {
        uint32_t v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = p[(c + q) * step];
#pragma unroll
        for (int q = 0; q < 8; ++q) sum += v[q];
    }
    for (; c < chunks; ++c) sum += p[c * step];
    hist[(uint64_t)s * LRB_HIST_BINS + b] = sum;
}

// ---------------------------------------------------------------------------
// K5: left-over read assignment (cluster_utils.py:261-268,309-322).  For read u and
// cluster c:  p = sum_f log( exp(-0.5 z^2) / (sqrt(2 pi) sigma) + 1e-7 ),
// z = (x - mu) / sigma, in float64 like numpy; a zero sigma makes p nan (0/0), nan
// never wins, the first maximum wins, best = -1 when every cluster is nan.
// One wave per read: lanes split the features, clusters are walked in order.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gauss_assign_kernel(const double *__restrict__ X,
                                                           uint64_t n_rows, int feats,
                                                           const double *__restrict__ mean,
                                                           const double *__restrict__ stdv,
                                                           int n_clusters, int32_t *__restrict__ best,
                                                           double *__restrict__ best_p)
{
    const uint32_t lane = lane_id();
    const uint64_t wave0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const double sqrt2pi = 2.5066282746310002; // np.sqrt(2*np.pi)
    for (uint64_t u = wave0; u < n_rows; u += nwaves) {
        const double *x = X + u * feats;
        double maxp = -__builtin_inf();
        int32_t arg = -1;
        for (int c = 0; c < n_clusters; ++c) {
            double part = 0.0;
            for (int f = lane; f < feats; f += WAVE) {
                const double sd = stdv[(size_t)c * feats + f];
                const double z = (x[f] - mean[(size_t)c * feats + f]) / sd;
                part += log(exp(-0.5 * (z * z)) / (sqrt2pi * sd) + 0.0000001);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, WAVE);
            if (part > maxp) { // false for nan
                maxp = part;
                arg = c;
            }
        }
        if (lane == 0) {
            best[u] = arg;
            if (best_p) best_p[u] = maxp;
        }
    }
}


// ===========================================================================
// C ABI
// ===========================================================================
// ---- K4 --------------------------------------------------------------------
extern "C" int lrb_seed_dist_dev(lrb_ctx *c, const float *d_M, uint64_t n_rows, int dims,
                                 uint64_t seed, float *d_out)
{
    ARG_TRY(c != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    ARG_TRY(dims >= 1 && dims <= 64);
    if (n_rows == 0) return LRB_OK;
    ARG_TRY(d_M && d_out && seed < n_rows);
    uint64_t blocks = (n_rows + 255) / 256;
    if (blocks > (uint64_t)c->n_cu * 8) blocks = (uint64_t)c->n_cu * 8;
    hipLaunchKernelGGL(seed_dist_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream, d_M,
                       n_rows, dims, seed, d_out);
    HIP_TRY(hipGetLastError());
    return LRB_OK;
}

template <int DIMS, int MAXD>
static void launch_seed_hist(lrb_ctx *c, dim3 grid, const float *d_M, uint64_t n, int dims,
                             const int64_t *d_seeds, uint32_t n_seeds, uint32_t chunk, uint32_t *d_hist)
{
    hipLaunchKernelGGL((seed_hist_kernel<DIMS, MAXD>), grid, dim3(256), 0, c->stream, d_M, n, dims,
                       d_seeds, n_seeds, chunk, d_hist);
}

extern "C" int lrb_seed_hist_dev(lrb_ctx *c, const float *d_M, uint64_t n_rows, int dims,
                                 const int64_t *d_seeds, uint32_t n_seeds, uint32_t *d_hist)
{
    ARG_TRY(c != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    ARG_TRY(dims >= 1 && dims <= 64);
    if (n_seeds == 0) return LRB_OK;
    ARG_TRY(d_hist != nullptr);
    if (n_rows == 0) {
        HIP_TRY(hipMemsetAsync(d_hist, 0, (size_t)n_seeds * LRB_HIST_BINS * 4, c->stream));
        return LRB_OK;
    }
    ARG_TRY(d_M && d_seeds);
    // 256 seeds a workgroup (y), the points cut into chunks (x) so that two workgroups a CU are there (61 KB of LDS each)
    const uint32_t sblocks = (n_seeds + SEEDS_PER_WG - 1) / SEEDS_PER_WG;
    uint64_t chunks = ((uint64_t)c->n_cu * 2 + sblocks - 1) / sblocks;
    if (chunks < 1) chunks = 1;
    uint64_t chunk = (n_rows + chunks - 1) / chunks;
    if (chunk < 256) chunk = 256; // (a workgroup's flush is 256 x 60 words: not for a handful of points)
    ARG_TRY(chunk <= 0xFFFFFFFFull);
    chunks = (n_rows + chunk - 1) / chunk;
    dim3 grid((unsigned)chunks, sblocks);
    void *d_part;
    int rc = lrb_ws_get(c, 17, chunks * LRB_HIST_BINS * (uint64_t)sblocks * SEEDS_PER_WG * sizeof(uint32_t), &d_part);
    if (rc != LRB_OK) return rc;
#define SEED_HIST_CASE(D, MD) \
    launch_seed_hist<D, MD>(c, grid, d_M, n_rows, dims, d_seeds, n_seeds, (uint32_t)chunk, (uint32_t *)d_part)
    switch (dims) {
    case 1: SEED_HIST_CASE(1, 1); break;
    case 2: SEED_HIST_CASE(2, 2); break;
    case 3: SEED_HIST_CASE(3, 3); break;
    case 4: SEED_HIST_CASE(4, 4); break;
    case 5: SEED_HIST_CASE(5, 5); break;
    case 6: SEED_HIST_CASE(6, 6); break;
    case 7: SEED_HIST_CASE(7, 7); break;
    case 8: SEED_HIST_CASE(8, 8); break;
    default:
        if (dims <= 16) SEED_HIST_CASE(0, 16);
        else if (dims <= 32) SEED_HIST_CASE(0, 32);
        else SEED_HIST_CASE(0, 64);
        break;
    }
#undef SEED_HIST_CASE
    hipLaunchKernelGGL(seed_hist_sum_kernel, dim3((n_seeds + 255) / 256, LRB_HIST_BINS), dim3(256), 0, c->stream, (const uint32_t *)d_part,
                       (uint32_t)chunks, sblocks * SEEDS_PER_WG, n_seeds, d_hist);
    HIP_TRY(hipGetLastError());
    return LRB_OK;
}

// ---- K5 --------------------------------------------------------------------
extern "C" int lrb_gauss_assign_dev(lrb_ctx *c, const double *d_X, uint64_t n_rows, int feats,
                                    const double *d_mean, const double *d_std, int n_clusters,
                                    int32_t *d_best, double *d_best_p)
{
    ARG_TRY(c != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    ARG_TRY(feats >= 1 && n_clusters >= 0);
    if (n_rows == 0) return LRB_OK;
    ARG_TRY(d_X && d_best && (n_clusters == 0 || (d_mean && d_std)));
    const int grid = grid_for_waves(c, n_rows, 4, 8);
    hipLaunchKernelGGL(gauss_assign_kernel, dim3(grid), dim3(256), 0, c->stream, d_X, n_rows, feats,
                       d_mean, d_std, n_clusters, d_best, d_best_p);
    HIP_TRY(hipGetLastError());
    return LRB_OK;
}

