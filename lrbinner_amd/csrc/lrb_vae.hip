// lrb_vae.hip -- the VAE training step of ae_utils.py (VAE.forward / calc_loss /
// trainepoch, ae_utils.py:163-241,243-271) as 12 fused fp32 kernels per step instead of the
// ~190 framework kernels the same step costs through autograd, gfx950 only.
//
// Why: the network is tiny (42-128-128-4-128-128-42 at the reference's test configuration,
// 46 k parameters, 0.28 GFLOP per 1024-row step) and the reference's schedule is 200 epochs
// of sequential 1024-row steps, so the step is bound by the NUMBER of kernels, not by
// arithmetic; on the whole pipeline it is 85 % of the wall time
// (DESIGN.md 3.6).  Every kernel here is a 16-row x 128-column tile GEMM on the matrix cores
// (v_mfma_f32_16x16x4_f32) with the surrounding element-wise work folded into its prologue /
// epilogue:
//
//   block  = BatchNorm(Dropout(LeakyReLU(Linear(x))))         ae_utils.py:130-133,173-176
//   fwd    : [BN of the previous block applied while loading] -> GEMM -> bias, LeakyReLU,
//            dropout, store, per-column sum / sum of squares (the batch statistics)
//   heads  : mu | logsigma in one GEMM, softplus, reparameterisation, KLD -- and, z being complete per row,
//            the first decoder block in the same kernel
//   out    : GEMM -> reconstruction error, loss terms, dL/drecon
//   bwd_dx : BatchNorm-backward + dropout + LeakyReLU' while loading dY -> dZ stored,
//            dX = dZ W, the two BatchNorm-backward sums of the block below (the first decoder layer's
//            instance also does the reparameterisation / KLD backward and the heads' dX)
//   bwd_dw : dW = dZ^T X over a slice of the batch, all layers in one launch (partials, summed
//            by the optimiser)
//   adam   : sums the partials, Adam, BatchNorm running statistics, next step's batch and counters
//
// The whole step is recorded once per batch size in a hipGraph; the only state that changes
// between replays (step number, position in the epoch's permutation) lives in device memory.
// Random numbers (dropout masks, eps) are a counter-based hash of (seed, step, layer,
// element), recomputed in the backward pass instead of stored.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "lrb_device.h"

#define VT_M 16   // rows of the batch per workgroup
#define VT_N 128  // output columns per chunk
#define VT_KC 64  // reduction chunk of the B operand held in LDS
#define VAE_DW_SMEM_MAX ((size_t)152 * 1024) // LDS one dW workgroup may take (160 KB a CU)
#define VT_NS 144 // row stride of that chunk: 144 % 64 = 16, so the 4 k-rows of one MFMA read hit 64 different banks
#define VAE_MAX_WIDTH 1024
#define VAE_BN_EPS 1e-5f
#define VAE_SLOPE 0.01f

struct vae_state {
    unsigned long long step;  // optimiser steps taken (Adam's t - 1)
    unsigned long long pos;   // offset of the current batch in the permutation
    unsigned long long limit; // rows of the permutation this call may touch
};

__device__ __forceinline__ uint32_t vae_hash(uint32_t seed, uint32_t step, uint32_t stream, uint32_t idx)
{
    uint32_t h = seed ^ (step * 0x9E3779B9u) ^ (stream * 0x85EBCA6Bu);
    uint32_t x = idx * 0x9E3779B1u + h;
    x ^= x >> 16;
    x *= 0x85EBCA6Bu;
    x ^= x >> 13;
    x *= 0xC2B2AE35u;
    x ^= x >> 16;
    return x;
}

__device__ __forceinline__ float vae_normal(uint32_t seed, uint32_t step, uint32_t stream, uint32_t idx)
{
    const uint32_t a = vae_hash(seed, step, stream, 2u * idx), b = vae_hash(seed, step, stream, 2u * idx + 1u);
    const float u1 = ((float)a + 1.0f) * 2.3283064365386963e-10f; // (0, 1]
    const float u2 = (float)b * 2.3283064365386963e-10f;
    return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}

// C[16 x 128 chunk] += As[16][kc] * Bs[kc][128] on the matrix cores (v_mfma_f32_16x16x4_f32, full
// fp32).  Wave w owns columns 32w..32w+31 as two 16x16 tiles; per 4 reduction steps a lane reads
// ONE float of A (row lane%16, step lane/16) and one of B per tile -- 3 LDS words per 2 MFMAs.
// Lane l holds acc[t][i] = C[row 4*(l/16) + i][column 32w + 16t + l%16].
// As rows are padded with zeros to a multiple of 4 columns, Bs rows past the chunk are zero.
typedef float v4f_t __attribute__((ext_vector_type(4)));

// NT = 16-column tiles per wave: 2 (a 128-column chunk per workgroup, above) or 1 -- a 64-column chunk: half the
// weights to stage and half the MFMAs per workgroup, for layers of at most 64 columns and for 128-column layers
// split over two workgroups while there are CUs to spare (a step's kernels are chains of latencies, not throughput).
template <int NT> struct vae_tile {
    static constexpr int N = 64 * NT;                 // output columns per chunk
    static constexpr int NS = NT == 2 ? VT_NS : 80;   // row stride of the B chunk in LDS: NS % 64 == 16 either way
    static constexpr int F4ROW = 16 * NT;             // float4 per chunk row
    static constexpr int F4 = VT_KC * F4ROW / 256;    // float4 per thread and chunk
};

template <int NT>
__device__ __forceinline__ void vae_tile_mfma(const float *As, int lda, int k0, const float *Bs, int kc, int lane,
                                              int wave, v4f_t (&acc)[NT])
{
    constexpr int NS = vae_tile<NT>::NS;
    const float *ap = As + (lane & 15) * lda + k0 + (lane >> 4);
    const float *bp = Bs + (lane >> 4) * NS + wave * (16 * NT) + (lane & 15);
    for (int k = 0; k < kc; k += 4) {
        const float a = ap[k];
        const float b0 = bp[k * NS];
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0, acc[0], 0, 0, 0);
        if (NT == 2) {
            const float b1 = bp[k * NS + 16];
            acc[NT - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1, acc[NT - 1], 0, 0, 0);
        }
    }
}

// output element j (0..4 NT - 1) of a lane: row / column inside the 16 x (64 NT) chunk
__device__ __forceinline__ int vae_orow(int lane, int j) { return (lane >> 4) * 4 + (j & 3); }
template <int NT = 2> __device__ __forceinline__ int vae_ocol(int lane, int wave, int j)
{
    return wave * (16 * NT) + (j >> 2) * 16 + (lane & 15);
}

// The B-operand chunks ([64 reduction rows][128 columns]) travel global -> registers -> LDS.
// A dependent global access costs 1.5-2 us on this part whatever its size (the previous
// kernel ran on other XCDs, so nothing is in this XCD's L2), and these kernels hold ~1 us of
// arithmetic: the only thing that matters is how many such round trips are chained.  So every
// kernel ISSUES all the loads it will need -- two chunks, the tile, the BatchNorm inputs --
// before it waits for any of them, and later chunks are fetched two iterations ahead.
template <int NT = 2> struct vae_wregs {
    float4 r[vae_tile<NT>::F4]; // 8 (4) x float4 per thread: float4 q = u * 256 + tid -> row q / 32 (16), columns 4 * (q % 32 (16))..
};

template <int NT, typename FetchF>
__device__ __forceinline__ void vae_wfetch(vae_wregs<NT> &w, int ch, int tid, FetchF fetch)
{
    constexpr int F4ROW = vae_tile<NT>::F4ROW;
#pragma unroll
    for (int u = 0; u < vae_tile<NT>::F4; ++u) {
        const int q = u * 256 + tid;
        w.r[u] = fetch(ch, q / F4ROW, (q % F4ROW) * 4);
    }
}

template <int NT, typename FixF>
__device__ __forceinline__ void vae_wstore(const vae_wregs<NT> &w, int ch, int tid, float *Bs, FixF fix)
{
    constexpr int F4ROW = vae_tile<NT>::F4ROW;
#pragma unroll
    for (int u = 0; u < vae_tile<NT>::F4; ++u) {
        const int q = u * 256 + tid;
        *reinterpret_cast<float4 *>(Bs + (q / F4ROW) * vae_tile<NT>::NS + (q % F4ROW) * 4) = fix(ch, q / F4ROW, (q % F4ROW) * 4, w.r[u]);
    }
}

// The B-operand chunks come through BUFFER loads: the hardware range check returns zeros past the end of the
// matrix, so the reduction rows beyond the last one need no predicate -- and a predicated global load is a branch,
// after which the compiler drains the memory counter: inside the chunk loop that turned the eight prefetches of
// a later chunk into eight chained round trips (layers wider than 128: the first and the output layer at k = 4, 5).
// Columns past the row end read into the next row; they only ever feed output columns that are not stored.
typedef uint32_t vae_v4u_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t vae_rsrc(const float *base, size_t n_floats)
{
    const uint64_t bytes = (uint64_t)n_floats * 4;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, bytes < 0x7FFFFFFFull ? (int)bytes : 0x7FFFFFFF, 0x00020000);
}

// AUX: cache-policy bits of the load: 0 as usual; 16 (sc1) bypasses the CU's L1 -- what the persistent XCD-local step
// reads with, where another CU of the XCD wrote the data during the same launch (the L1 is never refreshed by other CUs'
// stores; the XCD's L2 is their common ground)
template <int AUX = 0>
__device__ __forceinline__ float4 vae_bload4(__amdgpu_buffer_rsrc_t rs, uint32_t float_off)
{
    const vae_v4u_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(float_off * 4u), 0, AUX);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

template <int AUX = 0>
__device__ __forceinline__ float vae_bload1(__amdgpu_buffer_rsrc_t rs, uint32_t float_off)
{
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(float_off * 4u), 0, AUX));
}

// a plain load that another CU's store of this launch may have to be seen by (PX), else an ordinary one
template <bool PX, typename T> __device__ __forceinline__ T vae_ldg(const T *p)
{
    if (PX) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}

// A "virtual workgroup" of 256 threads: the bodies below are written against it, so that they can be walked by something
// other than one launch a layer -- round 3's persistent XCD-local step did (PX = true; measured slower, taken out in round
// 5: `git show fa9e017:lrbinner_amd/csrc/lrb_vae.hip`, profiles/r03_vae_px_probe.txt); the launches pass a real workgroup.
struct vae_vwg {
    int tid;            // 0..255
    int bx, by, nbx, nby;
    float *smem;        // this virtual workgroup's LDS
    float *wsum;        // [4][2] scratch
};

// four consecutive floats of a row: one 16-byte load when the row start and the offset allow it
__device__ __forceinline__ float4 vae_load4(const float *row, int c, int width, bool vec)
{
    if (vec) return *reinterpret_cast<const float4 *>(row + c);
    float4 v;
    v.x = c < width ? row[c] : 0.0f;
    v.y = c + 1 < width ? row[c + 1] : 0.0f;
    v.z = c + 2 < width ? row[c + 2] : 0.0f;
    v.w = c + 3 < width ? row[c + 3] : 0.0f;
    return v;
}

// per-column sums over the 16 rows of the tile: v[4t + i] -> lanes 0..15 get the sum of column
// (16 NT wave + 16 t + lane) in out[t]
template <int NT>
__device__ __forceinline__ void vae_col_sums(const float (&v)[4 * NT], float (&out)[NT])
{
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float sacc = v[4 * t] + v[4 * t + 1] + v[4 * t + 2] + v[4 * t + 3];
        sacc += __shfl_xor(sacc, 16, 64);
        sacc += __shfl_xor(sacc, 32, 64);
        out[t] = sacc;
    }
}

#define VAE_REPS 4 // copies of the per-step sums a step may spread its float atomics over
struct vae_bn {
    const float *stats; // [2][n]: sum, sum of squares of the block's output over the batch
    const float *gamma, *beta;
    // Large batches: hundreds of workgroups adding into the same 2n addresses queue in the memory-side atomic unit
    // (6.5 us of a 20 us kernel at batch 8192).  Workgroup w then adds into copy w % reps (copies rep_stride floats
    // apart), and readers add the copies up: four range-checked loads, of which the ones past copy reps - 1 read zero.
    // reps is a function of the batch size alone (vae_reps_for), so that the static descriptors need not know it.
    unsigned rep_stride;
};

__host__ __device__ __forceinline__ unsigned vae_reps_for(int B) { return B >= 4096 ? 4u : (B >= 2048 ? 2u : 1u); }

// A workgroup's share of two per-step sums (a column's sum and sum of squares, or the two BatchNorm-backward sums).
// Normally float atomics into copy bx % reps -- the order in which the row tiles arrive differs from run to run, and
// with it the last bit of the sums.  det (LRB_VAE_DETERMINISTIC=1): plain stores into the row tile's OWN row of a
// [row tiles][n_stats] table, det_shift floats from the sums; vae_det_reduce_kernel adds the rows up in order after
// the launch.  (A column of a row tile is written by exactly one lane of one workgroup.)
__device__ __forceinline__ void vae_stat_add(float *base, int det, long long det_shift, unsigned bx, unsigned reps, unsigned stride,
                                             int i1, float v1, int i2, float v2)
{
    if (det) {
        float *so = base + det_shift + (size_t)bx * stride;
        so[i1] = v1;
        so[i2] = v2;
    } else {
        float *so = base + (size_t)(bx % reps) * stride;
        atomicAdd(&so[i1], v1);
        atomicAdd(&so[i2], v2);
    }
}

// stats[k] = part[0][k] + part[1][k] + ... in that order (deterministic mode): one thread a sum
__global__ __launch_bounds__(256) void vae_det_reduce_kernel(const float *__restrict__ part, int tiles, size_t n_stats, float *__restrict__ stats)
{
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n_stats) return;
    float s = 0.0f;
    for (int t = 0; t < tiles; ++t) s += part[(size_t)t * n_stats + k];
    stats[k] = s;
}

// sum over the copies of one per-step sum: rs covers (reps - 1) * stride + len floats from the first copy
// MULTI is a compile-time property of the launch (batch >= 2048): small batches run instances that know one
// copy only -- four loads where one will do cost every kernel of a 1024-row step half a microsecond.
template <bool MULTI, int AUX = 0> __device__ __forceinline__ float vae_bsum4(__amdgpu_buffer_rsrc_t rs, uint32_t off, uint32_t stride)
{
    const float a = vae_bload1<AUX>(rs, off);
    if (!MULTI) return a;
    const float b = vae_bload1<AUX>(rs, off + stride), c = vae_bload1<AUX>(rs, off + 2 * stride), d = vae_bload1<AUX>(rs, off + 3 * stride);
    return (a + b) + (c + d);
}
// all VAE_REPS copies are cleared every step, so a reader may add all of them whatever the step used
template <bool MULTI> __device__ __forceinline__ __amdgpu_buffer_rsrc_t vae_rsrc_reps(const float *base, size_t len, unsigned stride)
{
    return vae_rsrc(base, len ? (size_t)(MULTI ? VAE_REPS - 1 : 0) * stride + len : 0);
}

// scale / shift of a BatchNorm for the columns tid, tid + 256, ... : inputs loaded by vae_bn_fetch
// (issued early), table written by vae_bn_table
struct vae_bn_regs {
    float s[4], q[4], g[4], b[4];
};

template <bool MULTI, int AUX = 0> __device__ __forceinline__ void vae_bn_fetch(vae_bn_regs &r, const vae_bn &bn, int n, int tid)
{
    // range-checked loads, no predicates (columns >= n get values nobody uses; no BatchNorm: zero records)
    const size_t cnt = bn.stats ? (size_t)n : 0;
    const __amdgpu_buffer_rsrc_t st = vae_rsrc_reps<MULTI>(bn.stats, 2 * cnt, bn.rep_stride), ga = vae_rsrc(bn.gamma, cnt),
                                 be = vae_rsrc(bn.beta, cnt);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t k = (uint32_t)(tid + u * 256);
        r.s[u] = vae_bsum4<MULTI, AUX>(st, k, bn.rep_stride);
        r.q[u] = vae_bsum4<MULTI, AUX>(st, (uint32_t)n + k, bn.rep_stride);
        r.g[u] = vae_bload1<AUX>(ga, k);
        r.b[u] = vae_bload1<AUX>(be, k);
    }
}

// coef[k] = gamma * rstd, coef[n + k] = beta - mean * gamma * rstd
__device__ __forceinline__ void vae_bn_table(const vae_bn_regs &r, int n, int tid, float invB, float *coef)
{
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int k = tid + u * 256;
        if (k < n) {
            const float mean = r.s[u] * invB;
            float var = r.q[u] * invB - mean * mean;
            var = var > 0.0f ? var : 0.0f;
            const float sc = rsqrtf(var + VAE_BN_EPS) * r.g[u];
            coef[k] = sc;
            coef[n + k] = r.b[u] - mean * sc;
        }
    }
}

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
enum { VAE_ACT_BLOCK = 0, VAE_ACT_HEADS = 1, VAE_ACT_LOSS = 2 };

struct vae_fwd_args {
    const float *in;           // [B][K] activations of the block below (or the gathered batch)
    vae_bn bn_in;              // stats == nullptr: input used as is
    const float *Wt, *bias;    // [K][N4] (the transposed, row-padded mirror of the layer's weight), [N]
    float *out;                // BLOCK: post-dropout activations [B][N]; HEADS: mu|logsigma [B][2L]
    float *stats_out;          // BLOCK: [2][N] (copy 0; see vae_bn)
    unsigned rep_stride;
    // HEADS
    float *z, *eps;            // [B][L]
    // HEADS, training: the first decoder block, whose input z is complete per row here
    // (Linear -> LeakyReLU -> Dropout, and its batch sums), so that it needs no launch of its own
    const float *nx_Wt, *nx_bias; // K-major mirror [L][nx_N4], bias [nx_N]
    float *nx_out, *nx_stats;     // [B][nx_N], [2][nx_N]; nx_out == nullptr: not fused
    int nx_N, nx_layer;
    // LOSS
    const float *data;         // the targets: the gathered batch [B][N]
    float *grad;               // dL/drecon [B][N]
    float *sums_part;          // [workgroups][4]: -, e_cov, e_comp, kld of this workgroup's rows (x 1/B)
    int cov_size;
    float w_cov, w_comp;
    // common
    const vae_state *state;
    int B, K, N, layer;
    uint32_t seed;
    uint32_t keep_threshold;   // dropout: keep iff hash >= threshold
    float keep_scale;
    int eval;                  // inference: bn_in.stats holds {mean, mean^2 + var} (count 1), no batch statistics out
    float *zero;               // first kernel of a step: the per-step sums of the NEXT step (other parity), to clear
    int zero_n;
    int det;                   // deterministic mode: the batch sums as plain stores per row tile (vae_stat_add)
    long long det_shift;
};

template <int ACT, bool MULTI, int NT, bool PX = false>
__device__ __forceinline__ void vae_fwd_body(vae_fwd_args a, const vae_vwg &vw)
{
    constexpr int AUX = PX ? 16 : 0;   // cache policy of the loads (vae_bload4)
    (void)AUX;
    constexpr int TN = vae_tile<NT>::N;
    float *const smem = vw.smem;
    const int K4 = (a.K + 3) & ~3, lda = K4 + 1;
    float *As = smem;                               // [16][K4+1], columns K..K4 zero
    float *Bs = As + ((VT_M * lda + 3) & ~3);       // [KC][TNS]
    float *coef = Bs + VT_KC * VT_NS;               // [2][K]: scale, shift of the BatchNorm below (the chunk is sized for the wider tile)
    float (*const wsum)[2] = reinterpret_cast<float (*)[2]>(vw.wsum);
    const int tid = vw.tid, lane = tid & 63, wave = tid >> 6;
    const int row0 = vw.bx * VT_M;
    const float invB = a.eval ? 1.0f : 1.0f / (float)a.B;
    if (a.zero)
        for (int i = vw.bx * 256 + tid; i < a.zero_n; i += vw.nbx * 256) a.zero[i] = 0.0f;
    // chunk ch: output columns n0 = (col0 + ch / nK) * 128, reduction rows k0 = (ch % nK) * 64.  A wide layer (N > 128)
    // may be launched with one workgroup per 128-column chunk (vw.nby): at small batches there are CUs to spare,
    // and the chunks of one row tile need nothing from each other.
    const int nK = (a.K + VT_KC - 1) / VT_KC;
    const int col0 = vw.nby > 1 ? (int)vw.by : 0;
    const int nchunks = (vw.nby > 1 ? 1 : (a.N + TN - 1) / TN) * nK;
    const int N4 = (a.N + 3) & ~3;
    const __amdgpu_buffer_rsrc_t wrs = vae_rsrc(a.Wt, (size_t)a.K * N4);
    auto wfetch = [&](int ch, int row, int col) {
        const int n0 = (col0 + ch / nK) * TN, k0 = (ch % nK) * VT_KC;
        // Bs[k][n] = W[n0+n][k0+k], 16 bytes at a time from the K-major mirror (rows padded to N4, zeros)
        return vae_bload4<AUX>(wrs, (uint32_t)((k0 + row) * N4 + n0 + col));
    };
    auto nofix = [](int, int, int, float4 v) { return v; };
    // ---- all the loads of the prologue, issued together -- in the order they are needed: the memory counter
    //      retires in order, so a wait for the tile also waits for everything issued before it.  The BatchNorm
    //      inputs and the tile first (the table and the LDS tile are built while the weights are still on their way),
    //      then the two weight chunks ----
    const unsigned reps = MULTI ? vae_reps_for(a.B) : 1u;
    vae_bn_regs bnr;
    vae_bn_fetch<MULTI, AUX>(bnr, a.bn_in, a.K, tid);
    // the tile: thread (rr = tid / 16, cq = tid % 16) takes columns 4 (cq + 16 u) .. +3 of row rr
    const int rr = tid >> 4, cq = tid & 15;
    const bool rowok = row0 + rr < a.B;
    // rows past the batch read as zero (range check); columns in [K, K4) meet zero rows of the B operand
    const __amdgpu_buffer_rsrc_t xrs = vae_rsrc(a.in, (size_t)a.B * a.K);
    const uint32_t xoff = (uint32_t)((row0 + rr) * a.K);
    float4 xv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) xv[u] = vae_bload4<AUX>(xrs, xoff + 4 * (cq + 16 * u));
    vae_wregs<NT> w0, w1;
    vae_wfetch(w0, 0, tid, wfetch);
    if (nchunks > 1) vae_wfetch(w1, 1, tid, wfetch);
    // fused first decoder block: column tid of its weight (the first 8 latent dimensions) and bias
    float nxw[8], nxb = 0.0f;
    const int nxN4 = (a.nx_N + 3) & ~3;
    if (ACT == VAE_ACT_HEADS) {
        const bool nx = a.nx_out != nullptr && tid < a.nx_N;
#pragma unroll
        for (int l = 0; l < 8; ++l) nxw[l] = (nx && l < (a.N >> 1)) ? vae_ldg<PX>(&a.nx_Wt[(size_t)l * nxN4 + tid]) : 0.0f;
        nxb = nx ? vae_ldg<PX>(&a.nx_bias[tid]) : 0.0f;
    }
    const uint32_t step = (uint32_t)vae_ldg<PX>(&a.state->step);
    // ---- tables, tile ----
    if (a.bn_in.stats) {
        vae_bn_table(bnr, a.K, tid, invB, coef);
        lrb_barrier();
    }
    auto put4 = [&](int k, float4 v) {
        float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (k + j < K4) {
                float t = e[j];
                if (a.bn_in.stats && rowok && k + j < a.K) t = fmaf(t, coef[k + j], coef[a.K + k + j]);
                As[rr * lda + k + j] = t;
            }
    };
#pragma unroll
    for (int u = 0; u < 2; ++u) put4(4 * (cq + 16 * u), xv[u]);
    for (int k = 4 * (cq + 32); k < K4; k += 64) // wide first layers (K > 128)
        put4(k, vae_bload4<AUX>(xrs, xoff + k));
    v4f_t acc[NT];
    float bias[4 * NT], target[4 * NT];
    const __amdgpu_buffer_rsrc_t brs = vae_rsrc(a.bias, (size_t)a.N),
                                 trs = vae_rsrc(a.data, ACT == VAE_ACT_LOSS ? (size_t)a.B * a.N : 0);
    float ec_total = 0.0f, ep_total = 0.0f;
    for (int ch = 0; ch < nchunks; ++ch) {
        const int n0 = (col0 + ch / nK) * TN, k0 = (ch % nK) * VT_KC;
        const int kc = a.K - k0 < VT_KC ? a.K - k0 : VT_KC;
        lrb_barrier(); // the tile is complete / the previous chunk has been multiplied
        if (ch & 1) {
            vae_wstore(w1, ch, tid, Bs, nofix);
            if (ch + 2 < nchunks) vae_wfetch(w1, ch + 2, tid, wfetch);
        } else {
            vae_wstore(w0, ch, tid, Bs, nofix);
            if (ch + 2 < nchunks) vae_wfetch(w0, ch + 2, tid, wfetch);
        }
        if (k0 == 0) { // what the epilogue of this column chunk will need
            acc[0] = acc[NT - 1] = v4f_t{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int j = 0; j < 4 * NT; ++j) {
                const int n = n0 + vae_ocol<NT>(lane, wave, j), b = row0 + vae_orow(lane, j);
                const bool ok = n < a.N && b < a.B;
                (void)ok;
                bias[j] = vae_bload1<AUX>(brs, (uint32_t)n);
                target[j] = ACT == VAE_ACT_LOSS ? vae_bload1<AUX>(trs, (uint32_t)(b * a.N + n)) : 0.0f;
            }
        }
        lrb_barrier();
        vae_tile_mfma<NT>(As, lda, k0, Bs, kc, lane, wave, acc);
        if (k0 + VT_KC < a.K) continue; // more of the reduction to come
        // ---- epilogue: arithmetic first, then the stores (a load or a branch between stores
        //      makes the compiler drain the memory counter every time) ----
        float s1[4 * NT], s2[4 * NT], outv[4 * NT];
#pragma unroll
        for (int j = 0; j < 4 * NT; ++j) {
            const int n = n0 + vae_ocol<NT>(lane, wave, j), b = row0 + vae_orow(lane, j);
            const bool ok = n < a.N && b < a.B;
            float v = acc[j >> 2][j & 3] + bias[j];
            if (ACT == VAE_ACT_BLOCK) {
                v = v > 0.0f ? v : VAE_SLOPE * v;
                const uint32_t h = vae_hash(a.seed, step, (uint32_t)a.layer, (uint32_t)(b * a.N + n));
                v = h >= a.keep_threshold ? v * a.keep_scale : 0.0f;
                outv[j] = v;
            } else if (ACT == VAE_ACT_HEADS) {
                outv[j] = v; // raw; finished below
            } else {
                const float d = v - target[j];
                const float w = n < a.cov_size ? a.w_cov : a.w_comp;
                outv[j] = 2.0f * w * d * invB;
                v = d * d; // for the loss sums
            }
            v = ok ? v : 0.0f;
            s1[j] = v;
            s2[j] = v * v;
        }
        {
            float *dst = ACT == VAE_ACT_LOSS ? a.grad : a.out;
#pragma unroll
            for (int j = 0; j < 4 * NT; ++j) {
                const int n = n0 + vae_ocol<NT>(lane, wave, j), b = row0 + vae_orow(lane, j);
                if (n < a.N && b < a.B) dst[(size_t)b * a.N + n] = outv[j];
                // the heads' second phase reads its tile back: from LDS when it fits behind the BatchNorm table
                // (of the 5 K floats the launch reserves there the forward kernel uses 2 K), else from memory
                if (ACT == VAE_ACT_HEADS && VT_M * a.N <= 3 * a.K && n < a.N) coef[2 * a.K + vae_orow(lane, j) * a.N + n] = outv[j];
            }
        }
        if (ACT == VAE_ACT_BLOCK) {
            if (!a.eval) {
                float c1[NT], c2[NT];
                vae_col_sums<NT>(s1, c1);
                vae_col_sums<NT>(s2, c2);
                if (lane < 16) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int n = n0 + wave * (16 * NT) + t * 16 + lane;
                        if (n < a.N) {
                            vae_stat_add(a.stats_out, a.det, a.det_shift, (unsigned)vw.bx, reps, a.rep_stride, n, c1[t], a.N + n, c2[t]);
                        }
                    }
                }
            }
        } else if (ACT == VAE_ACT_LOSS) {
            // squared error of this lane's 8 elements, split into the coverage / composition parts
#pragma unroll
            for (int j = 0; j < 4 * NT; ++j) {
                const int n = n0 + vae_ocol<NT>(lane, wave, j);
                if (n < a.cov_size) ec_total += s1[j];
                else ep_total += s1[j];
            }
        }
    }
    if (ACT == VAE_ACT_LOSS) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            ec_total += __shfl_xor(ec_total, o, 64);
            ep_total += __shfl_xor(ep_total, o, 64);
        }
        if (lane == 0) {
            wsum[wave][0] = ec_total;
            wsum[wave][1] = ep_total;
        }
        lrb_barrier();
        if (tid == 0) {
            const size_t wg = (size_t)vw.by * vw.nbx + vw.bx; // [column chunk][row tile]
            a.sums_part[wg * 4 + 1] = (wsum[0][0] + wsum[1][0] + wsum[2][0] + wsum[3][0]) * invB;
            a.sums_part[wg * 4 + 2] = (wsum[0][1] + wsum[1][1] + wsum[2][1] + wsum[3][1]) * invB;
        }
    }
    if (ACT == VAE_ACT_HEADS && !a.eval) {
        // out = [mu | raw logsigma]; softplus, reparameterise, KLD (ae_utils.py:135-139,163-170,259)
        const bool in_lds = VT_M * a.N <= 3 * a.K;
        const float *hs = coef + 2 * a.K; // [16][N]
        if (!in_lds) __threadfence_block();
        lrb_barrier();
        const int L = a.N >> 1;
        float *zs = Bs; // [16][L]: the GEMM is over, its B chunk is free (L <= 64 when fused)
        float kl = 0.0f;
        for (int i = tid; i < VT_M * L; i += 256) {
            const int rr = i / L, l = i - rr * L, bb = row0 + rr;
            if (bb < a.B) {
                const float mu = in_lds ? hs[rr * a.N + l] : vae_ldg<PX>(&a.out[(size_t)bb * a.N + l]);
                const float raw = in_lds ? hs[rr * a.N + L + l] : vae_ldg<PX>(&a.out[(size_t)bb * a.N + L + l]);
                const float ls = raw > 20.0f ? raw : log1pf(expf(raw));
                const float e = vae_normal(a.seed, step, (uint32_t)a.layer, (uint32_t)(bb * L + l));
                a.out[(size_t)bb * a.N + L + l] = ls;
                a.eps[(size_t)bb * L + l] = e;
                const float zz = mu + e * expf(0.5f * ls);
                a.z[(size_t)bb * L + l] = zz;
                if (a.nx_out) zs[i] = zz;
                kl += -0.5f * (1.0f + ls - mu * mu - expf(ls));
            } else if (a.nx_out) {
                zs[i] = 0.0f;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) kl += __shfl_xor(kl, o, 64);
        if (lane == 0) wsum[wave][0] = kl;
        lrb_barrier();
        if (tid == 0) a.sums_part[vw.bx * 4 + 3] = (wsum[0][0] + wsum[1][0] + wsum[2][0] + wsum[3][0]) * invB;
        if (a.nx_out) {
            // ---- the first decoder block on this workgroup's 16 rows of z (reduction length L: plain FMAs,
            //      one output column per thread, so the column sums are thread-local) ----
            const int rows = a.B - row0 < VT_M ? a.B - row0 : VT_M;
            for (int n = tid; n < a.nx_N; n += 256) {
                float o[VT_M];
                const float bv = n == tid ? nxb : vae_ldg<PX>(&a.nx_bias[n]);
#pragma unroll
                for (int r = 0; r < VT_M; ++r) o[r] = bv;
#pragma unroll
                for (int l = 0; l < 8; ++l)
                    if (l < L) {
                        const float w = n == tid ? nxw[l] : vae_ldg<PX>(&a.nx_Wt[(size_t)l * nxN4 + n]);
#pragma unroll
                        for (int r = 0; r < VT_M; ++r) o[r] = fmaf(zs[r * L + l], w, o[r]);
                    }
                for (int l = 8; l < L; ++l) {
                    const float w = vae_ldg<PX>(&a.nx_Wt[(size_t)l * nxN4 + n]);
#pragma unroll
                    for (int r = 0; r < VT_M; ++r) o[r] = fmaf(zs[r * L + l], w, o[r]);
                }
                float c1 = 0.0f, c2 = 0.0f;
#pragma unroll
                for (int r = 0; r < VT_M; ++r) {
                    float v = o[r];
                    v = v > 0.0f ? v : VAE_SLOPE * v;
                    const uint32_t h = vae_hash(a.seed, step, (uint32_t)a.nx_layer, (uint32_t)((row0 + r) * a.nx_N + n));
                    v = h >= a.keep_threshold ? v * a.keep_scale : 0.0f;
                    v = r < rows ? v : 0.0f;
                    o[r] = v;
                    c1 += v;
                    c2 += v * v;
                }
#pragma unroll
                for (int r = 0; r < VT_M; ++r)
                    if (r < rows) a.nx_out[(size_t)(row0 + r) * a.nx_N + n] = o[r];
                vae_stat_add(a.nx_stats, a.det, a.det_shift, (unsigned)vw.bx, reps, a.rep_stride, n, c1, a.nx_N + n, c2);
            }
        }
    }
}

template <int ACT, bool MULTI, int NT>
__global__ __launch_bounds__(256) void vae_fwd_kernel(vae_fwd_args a)
{
    extern __shared__ __attribute__((aligned(16))) float smem_dyn[];
    __shared__ float wsum_st[4][2];
    const vae_vwg vw{(int)threadIdx.x, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x, (int)gridDim.y, smem_dyn, &wsum_st[0][0]};
    vae_fwd_body<ACT, MULTI, NT, false>(a, vw);
}

// ---------------------------------------------------------------------------
// backward: dZ from dY (prologue), dX = dZ W, BatchNorm-backward sums of the block below
// ---------------------------------------------------------------------------
struct vae_bwd_args {
    const float *dY;        // [B][N] gradient w.r.t. this layer's output (BLOCK: w.r.t. the BN output)
    const float *act;       // BLOCK: the block's stored post-dropout activations [B][N]
    vae_bn bn;              // BLOCK: this block's BatchNorm
    const float *bsum;      // BLOCK: [2][N] sum dY, sum dY * xhat
    float *dZ;              // BLOCK: [B][N] written (gradient w.r.t. the Linear output)
    const float *W;         // [N][K4]: the row-padded copy of the layer's weight
    float *dX;              // [B][K] or nullptr (first layer)
    const float *act_below; // activations of the block below [B][K] (for its xhat), or nullptr
    vae_bn bn_below;
    float *bsum_below;      // [2][K]
    const vae_state *state;
    int B, K, N, layer, block;
    uint32_t seed, keep_threshold;
    float keep_scale;
    // first decoder layer: dX is d(z); the reparameterisation / KLD backward follows in place
    // (ae_utils.py:163-170,259) and d(mu | raw logsigma) [B][2K] is what gets stored
    const float *heads, *eps;
    float *dheads;
    float w_kld;
    // ... and, fused, the backward of the two heads: dY of the last encoder block = dheads W_heads (reduction
    // length 2K), with that block's BatchNorm-backward sums
    const float *h_W;         // [2K][h_K4] row-padded copy of [mu.W; ls.W]; nullptr: not fused
    float *h_dX;              // [B][h_K]
    const float *h_act_below; // [B][h_K]
    vae_bn h_bn_below;
    float *h_bsum_below;      // [2][h_K]
    unsigned rep_stride;      // between the copies of the per-step sums (vae_bn)
    int h_K;
    int det;                  // deterministic mode (vae_stat_add)
    long long det_shift;
};

template <bool LATENT, bool MULTI, int NT, bool PX = false>
__device__ __forceinline__ void vae_bwd_dx_body(vae_bwd_args a, const vae_vwg &vw)
{
    constexpr int AUX = PX ? 16 : 0;   // cache policy of the loads (vae_bload4)
    (void)AUX;
    constexpr int TN = vae_tile<NT>::N;
    float *const smem = vw.smem;
    const int N4 = (a.N + 3) & ~3, lda = N4 + 1;
    float *As = smem;                         // dZ tile [16][N4+1], columns N..N4 zero
    float *Bs = As + ((VT_M * lda + 3) & ~3); // [KC][VT_NS]
    float *cn = Bs + VT_KC * VT_NS;           // [5][N]: mean, rstd, gamma*rstd, S1/B, S2/B of this block
    float *ck = cn + 5 * a.N;                 // [2][K]: mean, rstd of the block below
    const int tid = vw.tid, lane = tid & 63, wave = tid >> 6;
    const int row0 = vw.bx * VT_M;
    const float invB = 1.0f / (float)a.B;
    // chunk ch: output columns k0 = (col0 + ch / nR) * TN, reduction rows n0 = (ch % nR) * 64; with vw.nby > 1 a
    // workgroup owns ONE column chunk of the row tile (every one of them builds the dZ tile; the first stores it)
    const int col0 = vw.nby > 1 ? (int)vw.by : 0;
    const int nR = (a.N + VT_KC - 1) / VT_KC, nchunks = a.dX ? (vw.nby > 1 ? 1 : (a.K + TN - 1) / TN) * nR : 0;
    const int K4 = (a.K + 3) & ~3;
    const __amdgpu_buffer_rsrc_t wrs = vae_rsrc(a.W, (size_t)a.N * K4);
    auto wfetch = [&](int ch, int row, int col) {
        const int k0 = (col0 + ch / nR) * TN, n0 = (ch % nR) * VT_KC;
        // Bs[n][k] = W[n0+n][k0+k] from the row-padded copy
        return vae_bload4<AUX>(wrs, (uint32_t)((n0 + row) * K4 + k0 + col));
    };
    auto nofix = [](int, int, int, float4 v) { return v; };
    // ---- all the loads of the prologue, issued together, in the order they are needed (see the forward kernel):
    //      the table inputs and the dY / activation tiles first, the weight chunks last ----
    vae_wregs<NT> w0, w1;
    // range-checked loads, no predicates: what lies past a matrix reads as zero, what lies past a row end is
    // masked where it is used
    float t_s[4], t_q[4], t_g[4], t_1[4], t_2[4], k_s[4], k_q[4];
    {
        const size_t cn_ = a.block ? (size_t)a.N : 0, ck_ = (a.bsum_below && a.dX) ? (size_t)a.K : 0;
        const unsigned rstr = a.rep_stride;
        const __amdgpu_buffer_rsrc_t st = vae_rsrc_reps<MULTI>(a.bn.stats, 2 * cn_, rstr), ga = vae_rsrc(a.bn.gamma, cn_),
                                     bs = vae_rsrc_reps<MULTI>(a.bsum, 2 * cn_, rstr), sk = vae_rsrc_reps<MULTI>(a.bn_below.stats, 2 * ck_, rstr);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t n = (uint32_t)(tid + u * 256);
            t_s[u] = vae_bsum4<MULTI, AUX>(st, n, rstr);
            t_q[u] = vae_bsum4<MULTI, AUX>(st, (uint32_t)a.N + n, rstr);
            t_g[u] = vae_bload1<AUX>(ga, n);
            t_1[u] = vae_bsum4<MULTI, AUX>(bs, n, rstr);
            t_2[u] = vae_bsum4<MULTI, AUX>(bs, (uint32_t)a.N + n, rstr);
            k_s[u] = vae_bsum4<MULTI, AUX>(sk, n, rstr);
            k_q[u] = vae_bsum4<MULTI, AUX>(sk, (uint32_t)a.K + n, rstr);
        }
    }
    // the dY / activation tiles: thread (rr = tid / 16, cq = tid % 16), columns 4 (cq + 16 u) .. +3
    const int rr = tid >> 4, cq = tid & 15;
    const bool rowok = row0 + rr < a.B;
    const __amdgpu_buffer_rsrc_t yrs = vae_rsrc(a.dY, (size_t)a.B * a.N), ars = vae_rsrc(a.act, a.block ? (size_t)a.B * a.N : 0);
    const uint32_t yoff = (uint32_t)((row0 + rr) * a.N);
    float4 gv[2], dv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        gv[u] = vae_bload4<AUX>(yrs, yoff + 4 * (cq + 16 * u));
        dv[u] = vae_bload4<AUX>(ars, yoff + 4 * (cq + 16 * u));
    }
    if (nchunks > 0) vae_wfetch(w0, 0, tid, wfetch);
    if (nchunks > 1) vae_wfetch(w1, 1, tid, wfetch);
    float hw[16], h_s = 0.0f, h_q = 0.0f; // fused heads backward: column tid of W_heads (first 16 rows), BN sums below
    const int HK4 = (a.h_K + 3) & ~3;
    {
        const bool hx = LATENT && a.h_W != nullptr && tid < a.h_K;
#pragma unroll
        for (int c = 0; c < 16; ++c) hw[c] = (hx && c < 2 * a.K) ? vae_ldg<PX>(&a.h_W[(size_t)c * HK4 + tid]) : 0.0f;
        if (hx) {
            const __amdgpu_buffer_rsrc_t hs_ = vae_rsrc_reps<MULTI>(a.h_bn_below.stats, 2 * (size_t)a.h_K, a.rep_stride);
            h_s = vae_bsum4<MULTI, AUX>(hs_, (uint32_t)tid, a.rep_stride);
            h_q = vae_bsum4<MULTI, AUX>(hs_, (uint32_t)(a.h_K + tid), a.rep_stride);
        }
    }
    const uint32_t step = (uint32_t)vae_ldg<PX>(&a.state->step);
    // ---- tables ----
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int n = tid + u * 256;
        if (a.block && n < a.N) {
            const float mean = t_s[u] * invB;
            float var = t_q[u] * invB - mean * mean;
            var = var > 0.0f ? var : 0.0f;
            const float rstd = rsqrtf(var + VAE_BN_EPS);
            cn[n] = mean;
            cn[a.N + n] = rstd;
            cn[2 * a.N + n] = t_g[u] * rstd;
            cn[3 * a.N + n] = t_1[u] * invB;
            cn[4 * a.N + n] = t_2[u] * invB;
        }
        if (a.bsum_below && a.dX && n < a.K) {
            const float mean = k_s[u] * invB;
            float var = k_q[u] * invB - mean * mean;
            var = var > 0.0f ? var : 0.0f;
            ck[n] = mean;
            ck[a.K + n] = rsqrtf(var + VAE_BN_EPS);
        }
    }
    if (tid < VT_M)
        for (int n = a.N; n < N4; ++n) As[tid * lda + n] = 0.0f;
    lrb_barrier();
    // ---- dZ tile: BatchNorm backward, dropout, LeakyReLU' ----
    auto put4 = [&](int n, float4 g4, float4 d4) {
        float g[4] = {g4.x, g4.y, g4.z, g4.w};
        const float d[4] = {d4.x, d4.y, d4.z, d4.w};
        const int b = row0 + rr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nn = n + j < a.N ? n + j : 0;
            if (a.block) {
                const float xhat = (d[j] - cn[nn]) * cn[a.N + nn];
                float t = cn[2 * a.N + nn] * (g[j] - cn[3 * a.N + nn] - xhat * cn[4 * a.N + nn]);
                const uint32_t h = vae_hash(a.seed, step, (uint32_t)a.layer, (uint32_t)(b * a.N + nn));
                t = h >= a.keep_threshold ? t * a.keep_scale : 0.0f;
                t = d[j] > 0.0f ? t : VAE_SLOPE * t;
                g[j] = t;
            }
            if (!rowok || n + j >= a.N) g[j] = 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (n + j < a.N) {
                As[rr * lda + n + j] = g[j];
                if (a.block && rowok && vw.by == 0) a.dZ[(size_t)b * a.N + n + j] = g[j];
            }
    };
#pragma unroll
    for (int u = 0; u < 2; ++u) put4(4 * (cq + 16 * u), gv[u], dv[u]);
    for (int n = 4 * (cq + 32); n < a.N; n += 64) { // layers wider than 128
        put4(n, vae_bload4<AUX>(yrs, yoff + n), vae_bload4<AUX>(ars, yoff + n));
    }
    if (!a.dX) return;
    v4f_t acc[NT];
    float below[4 * NT];
    const __amdgpu_buffer_rsrc_t lrs = vae_rsrc(a.act_below, a.bsum_below ? (size_t)a.B * a.K : 0);
    for (int ch = 0; ch < nchunks; ++ch) {
        const int k0 = (col0 + ch / nR) * TN, n0 = (ch % nR) * VT_KC; // output columns = inputs of the layer
        const int nc = a.N - n0 < VT_KC ? a.N - n0 : VT_KC;
        lrb_barrier();
        if (ch & 1) {
            vae_wstore(w1, ch, tid, Bs, nofix);
            if (ch + 2 < nchunks) vae_wfetch(w1, ch + 2, tid, wfetch);
        } else {
            vae_wstore(w0, ch, tid, Bs, nofix);
            if (ch + 2 < nchunks) vae_wfetch(w0, ch + 2, tid, wfetch);
        }
        if (n0 == 0) {
            acc[0] = acc[NT - 1] = v4f_t{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int j = 0; j < 4 * NT; ++j) {
                const int k = k0 + vae_ocol<NT>(lane, wave, j), b = row0 + vae_orow(lane, j);
                below[j] = vae_bload1<AUX>(lrs, (uint32_t)(b * a.K + k));
            }
        }
        lrb_barrier();
        vae_tile_mfma<NT>(As, lda, n0, Bs, nc, lane, wave, acc);
        if (n0 + VT_KC < a.N) continue;
        float s1[4 * NT], s2[4 * NT];
#pragma unroll
        for (int j = 0; j < 4 * NT; ++j) {
            const int k = k0 + vae_ocol<NT>(lane, wave, j), b = row0 + vae_orow(lane, j);
            const bool ok = k < a.K && b < a.B;
            const float g = ok ? acc[j >> 2][j & 3] : 0.0f;
            s1[j] = g;
            s2[j] = (a.bsum_below && ok) ? g * (below[j] - ck[k]) * ck[a.K + k] : 0.0f;
        }
        if (LATENT && a.dheads) {
            float mu[4 * NT], ls[4 * NT], ep[4 * NT];
#pragma unroll
            for (int j = 0; j < 4 * NT; ++j) {
                const int k = k0 + vae_ocol<NT>(lane, wave, j), b = row0 + vae_orow(lane, j);
                const bool ok = k < a.K && b < a.B;
                mu[j] = ok ? vae_ldg<PX>(&a.heads[(size_t)b * 2 * a.K + k]) : 0.0f;
                ls[j] = ok ? vae_ldg<PX>(&a.heads[(size_t)b * 2 * a.K + a.K + k]) : 0.0f;
                ep[j] = ok ? vae_ldg<PX>(&a.eps[(size_t)b * a.K + k]) : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < 4 * NT; ++j) {
                const float g = s1[j], sd = expf(0.5f * ls[j]);
                s2[j] = (g * ep[j] * 0.5f * sd + a.w_kld * (-0.5f) * (1.0f - expf(ls[j])) * invB) *
                        (1.0f - expf(-ls[j]));                    // d(raw logsigma): softplus' = 1 - exp(-softplus)
                s1[j] = g + a.w_kld * mu[j] * invB;               // d(mu)
            }
#pragma unroll
            for (int j = 0; j < 4 * NT; ++j) {
                const int k = k0 + vae_ocol<NT>(lane, wave, j), b = row0 + vae_orow(lane, j);
                if (k < a.K && b < a.B) {
                    a.dheads[(size_t)b * 2 * a.K + k] = s1[j];
                    a.dheads[(size_t)b * 2 * a.K + a.K + k] = s2[j];
                }
            }
            if (a.h_W) {
                lrb_barrier(); // every wave is through with the B chunk: it becomes the dheads tile [16][2K]
#pragma unroll
                for (int j = 0; j < 4 * NT; ++j) {
                    const int k = k0 + vae_ocol<NT>(lane, wave, j), r = vae_orow(lane, j);
                    if (k < a.K) {
                        Bs[r * 2 * a.K + k] = s1[j];
                        Bs[r * 2 * a.K + a.K + k] = s2[j];
                    }
                }
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < 4 * NT; ++j) {
            const int k = k0 + vae_ocol<NT>(lane, wave, j), b = row0 + vae_orow(lane, j);
            if (k < a.K && b < a.B) a.dX[(size_t)b * a.K + k] = s1[j];
        }
        if (a.bsum_below) {
            float c1[NT], c2[NT];
            vae_col_sums<NT>(s1, c1);
            vae_col_sums<NT>(s2, c2);
            if (lane < 16) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int k = k0 + wave * (16 * NT) + t * 16 + lane;
                    if (k < a.K) {
                        vae_stat_add(a.bsum_below, a.det, a.det_shift, (unsigned)vw.bx, MULTI ? vae_reps_for(a.B) : 1u, a.rep_stride, k, c1[t],
                                     a.K + k, c2[t]);
                    }
                }
            }
        }
    }
    if (LATENT && a.dheads && a.h_W) {
        // ---- heads backward on the tile (one output column per thread: the sums over rows are thread-local) ----
        lrb_barrier();
        const int C = 2 * a.K, rows = a.B - row0 < VT_M ? a.B - row0 : VT_M;
        const float *hs = Bs;
        for (int k = tid; k < a.h_K; k += 256) {
            float g[VT_M];
#pragma unroll
            for (int r = 0; r < VT_M; ++r) g[r] = 0.0f;
#pragma unroll
            for (int c = 0; c < 16; ++c)
                if (c < C) {
                    const float w = k == tid ? hw[c] : vae_ldg<PX>(&a.h_W[(size_t)c * HK4 + k]);
#pragma unroll
                    for (int r = 0; r < VT_M; ++r) g[r] = fmaf(hs[r * C + c], w, g[r]);
                }
            for (int c = 16; c < C; ++c) {
                const float w = vae_ldg<PX>(&a.h_W[(size_t)c * HK4 + k]);
#pragma unroll
                for (int r = 0; r < VT_M; ++r) g[r] = fmaf(hs[r * C + c], w, g[r]);
            }
            float sum = h_s, sq = h_q;
            if (k != tid) {
                const __amdgpu_buffer_rsrc_t hs_ = vae_rsrc_reps<MULTI>(a.h_bn_below.stats, 2 * (size_t)a.h_K, a.rep_stride);
                sum = vae_bsum4<MULTI, AUX>(hs_, (uint32_t)k, a.rep_stride);
                sq = vae_bsum4<MULTI, AUX>(hs_, (uint32_t)(a.h_K + k), a.rep_stride);
            }
            const float mean = sum * invB;
            float var = sq * invB - mean * mean;
            var = var > 0.0f ? var : 0.0f;
            const float rstd = rsqrtf(var + VAE_BN_EPS);
            float below[VT_M];
#pragma unroll
            for (int r = 0; r < VT_M; ++r) below[r] = r < rows ? vae_ldg<PX>(&a.h_act_below[(size_t)(row0 + r) * a.h_K + k]) : 0.0f;
            float c1 = 0.0f, c2 = 0.0f;
#pragma unroll
            for (int r = 0; r < VT_M; ++r)
                if (r < rows) {
                    c1 += g[r];
                    c2 += g[r] * (below[r] - mean) * rstd;
                }
#pragma unroll
            for (int r = 0; r < VT_M; ++r)
                if (r < rows) a.h_dX[(size_t)(row0 + r) * a.h_K + k] = g[r];
            vae_stat_add(a.h_bsum_below, a.det, a.det_shift, (unsigned)vw.bx, MULTI ? vae_reps_for(a.B) : 1u, a.rep_stride, k, c1, a.h_K + k, c2);
        }
    }
}

template <bool LATENT, bool MULTI, int NT>
__global__ __launch_bounds__(256) void vae_bwd_dx_kernel(vae_bwd_args a)
{
    extern __shared__ __attribute__((aligned(16))) float smem_dyn[];
    __shared__ float wsum_st[4][2];
    const vae_vwg vw{(int)threadIdx.x, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x, (int)gridDim.y, smem_dyn, &wsum_st[0][0]};
    vae_bwd_dx_body<LATENT, MULTI, NT, false>(a, vw);
}

// dW[n][k] = sum_b dZ[b][n] * X[b][k] over the rows of one slice of the batch; db likewise.
// ONE launch for all layers, after the dX chain: grid = (sum over layers of ceil(N / 16), slices);
// partial results go to part[slice][...] and are summed by the optimiser (deterministic).
struct vae_dw_desc {
    const float *dZ;        // [B][N]
    const float *in;        // [B][K] activations below (or the gathered batch)
    vae_bn bn_in;
    unsigned long long w_off, b_off; // this layer's dW / db inside a partial
    int K, N, tile0;        // tile0: first blockIdx.x of this layer
    // dZ made HERE instead of read (the first encoder block: nothing else wants its dZ, and a launch that only turns
    // dY into dZ costs what every launch costs, 4.7 us of the 93): dY = gradient w.r.t. the block's BatchNorm output,
    // act = its stored post-dropout activations, bn / bsum as vae_bwd_args; dY == nullptr: dZ is read
    const float *dY, *act;
    vae_bn bn;
    const float *bsum;
    int layer;
};

struct vae_dw_args {
    const float *dZ;
    const float *in;
    vae_bn bn_in;
    float *part;
    size_t n_params, w_off, b_off;
    int B, K, N, rows_per_slice;
};

template <bool MULTI, bool PX = false, bool EARLY = false>   // EARLY: the BatchNorm table first (below)
__device__ __forceinline__ void vae_bwd_dw_body(const vae_dw_desc *__restrict__ descs, int n_layers, float *part_all,
                                                         size_t n_params, int B, int rows_per_slice, const vae_state *state,
                                                         uint32_t seed, uint32_t keep_threshold, float keep_scale, const vae_vwg &vw)
{
    constexpr int AUX = PX ? 16 : 0;   // cache policy of the loads (vae_bload4)
    (void)AUX;
    int li = 0;
    while (li + 1 < n_layers && (int)vw.bx >= descs[li + 1].tile0) ++li;
    const vae_dw_desc d = descs[li];
    vae_dw_args a;
    a.dZ = d.dZ; a.in = d.in; a.bn_in = d.bn_in; a.part = part_all; a.n_params = n_params;
    a.w_off = d.w_off; a.b_off = d.b_off; a.B = B; a.K = d.K; a.N = d.N; a.rows_per_slice = rows_per_slice;
    const int tile_x = (int)vw.bx - d.tile0;
    float *const smem = vw.smem;
    // As[16 n][rows+1] = dZ^T tile, Bs[KC rows][VT_NS], coef[2][K]
    const int rows = a.rows_per_slice, lda = rows + 1; // rows is a multiple of 4
    float *As = smem;
    float *Bs = As + ((VT_M * lda + 3) & ~3);
    float *coef = Bs + VT_KC * VT_NS;
    const int tid = vw.tid, r = tid >> 4, c = tid & 15, lane = tid & 63, wave = tid >> 6;
    const int n0 = tile_x * VT_M;
    const int b0 = vw.by * rows;
    const float invB = 1.0f / (float)a.B;
    float *part = a.part + (size_t)vw.by * a.n_params;
    // chunk ch: output columns k0 = (ch / nR) * 128, batch rows r0 = (ch % nR) * 64 of this slice
    const int nR = (rows + VT_KC - 1) / VT_KC, nchunks = ((a.K + VT_N - 1) / VT_N) * nR;
    // X[b][k]: rows past the batch read as zero (range check); a row of this slice that is past the slice
    // (rows < 64 r) or a column past K meets a zero row of dZ^T / an output column that is not stored
    const __amdgpu_buffer_rsrc_t xrs = vae_rsrc(a.in, (size_t)a.B * a.K);
    auto wfetch = [&](int ch, int row, int col) {
        const int k0 = (ch / nR) * VT_N, r0 = (ch % nR) * VT_KC;
        return vae_bload4<AUX>(xrs, (uint32_t)((b0 + r0 + row) * a.K + k0 + col));
    };
    // The BatchNorm of the layer below is affine per input column, X' = sc[k] X + sh[k], so it moves out of the
    // reduction: dW[n][k] = sum_b dZ[b][n] X'[b][k] = sc[k] * (sum_b dZ[b][n] X[b][k]) + sh[k] * (sum_b dZ[b][n]), and
    // the second sum is this slice's bias gradient.  The raw activations go to LDS untouched (fixing every element
    // on the way in was 2 us per chunk: eight table reads and four range checks per 16 bytes).
    auto nofix = [](int, int, int, float4 v) { return v; };
    float *dbs = coef + 2 * a.K; // [16]: the slice's bias gradient per tile row
    // ---- all the loads of the prologue, issued together, in the order they are needed: the dZ^T tile (it goes to
    //      LDS first), the two chunks of activations, the BatchNorm inputs (used in the epilogue only) ----
    // dZ^T tile: thread (bb = tid / 4, c4 = tid % 4) takes columns n0 + 4 c4 .. +3 of rows bb, bb + 64, ...
    const bool make_dz = d.dY != nullptr;
    const __amdgpu_buffer_rsrc_t zrs = vae_rsrc(make_dz ? d.dY : a.dZ, (size_t)a.B * a.N),
                                 ars = vae_rsrc(d.act, make_dz ? (size_t)a.B * a.N : 0);
    const int bb0 = tid >> 2, c4 = (tid & 3) * 4;
    // make_dz: BatchNorm backward, dropout, LeakyReLU' of this thread's four columns (vae_bwd_dx_kernel's put4)
    float zt[4][5]; // mean, rstd, gamma * rstd, S1 / B, S2 / B
    uint32_t zstep = 0;
    if (make_dz) {
        const size_t cn_ = (size_t)a.N;
        const __amdgpu_buffer_rsrc_t st = vae_rsrc_reps<MULTI>(d.bn.stats, 2 * cn_, d.bn.rep_stride), ga = vae_rsrc(d.bn.gamma, cn_),
                                     bs = vae_rsrc_reps<MULTI>(d.bsum, 2 * cn_, d.bn.rep_stride);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t n = (uint32_t)(n0 + c4 + j);
            const float mean = vae_bsum4<MULTI, AUX>(st, n, d.bn.rep_stride) * invB;
            float var = vae_bsum4<MULTI, AUX>(st, (uint32_t)a.N + n, d.bn.rep_stride) * invB - mean * mean;
            var = var > 0.0f ? var : 0.0f;
            const float rstd = rsqrtf(var + VAE_BN_EPS);
            zt[j][0] = mean;
            zt[j][1] = rstd;
            zt[j][2] = vae_bload1<AUX>(ga, n) * rstd;
            zt[j][3] = vae_bsum4<MULTI, AUX>(bs, n, d.bn.rep_stride) * invB;
            zt[j][4] = vae_bsum4<MULTI, AUX>(bs, (uint32_t)a.N + n, d.bn.rep_stride) * invB;
        }
        zstep = (uint32_t)vae_ldg<PX>(&state->step);
    }
    auto zload = [&](int bb, float4 (&zv)[2]) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int b = b0 + bb + 64 * u;
            // rows past the batch read as zero; columns past N give output rows that are not stored
            zv[u] = bb + 64 * u < rows ? vae_bload4<AUX>(zrs, (uint32_t)(b * a.N + n0 + c4)) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (make_dz) {
                const float4 a4 = bb + 64 * u < rows ? vae_bload4<AUX>(ars, (uint32_t)(b * a.N + n0 + c4)) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                float g[4] = {zv[u].x, zv[u].y, zv[u].z, zv[u].w};
                const float dd[4] = {a4.x, a4.y, a4.z, a4.w};
                const bool rowok = bb + 64 * u < rows && b < a.B;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = n0 + c4 + j;
                    const float xhat = (dd[j] - zt[j][0]) * zt[j][1];
                    float t = zt[j][2] * (g[j] - zt[j][3] - xhat * zt[j][4]);
                    const uint32_t h = vae_hash(seed, zstep, (uint32_t)d.layer, (uint32_t)(b * a.N + n));
                    t = h >= keep_threshold ? t * keep_scale : 0.0f;
                    t = dd[j] > 0.0f ? t : VAE_SLOPE * t;
                    g[j] = (rowok && n < a.N) ? t : 0.0f;
                }
                zv[u] = make_float4(g[0], g[1], g[2], g[3]);
            }
        }
    };
    auto zstore = [&](int bb, const float4 (&zv)[2]) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (bb + 64 * u < rows) {
                As[(c4 + 0) * lda + bb + 64 * u] = zv[u].x;
                As[(c4 + 1) * lda + bb + 64 * u] = zv[u].y;
                As[(c4 + 2) * lda + bb + 64 * u] = zv[u].z;
                As[(c4 + 3) * lda + bb + 64 * u] = zv[u].w;
            }
    };
    float4 z0[2];
    zload(bb0, z0);
    vae_wregs w0, w1;
    vae_bn_regs bnr;
    if (EARLY) {
        // the BatchNorm table of the layer below BEFORE the two chunks of activations are asked for: with the replica
        // sums of the large-batch form its 40 loads, in flight beside the 64 registers of the two chunks, were six
        // registers more than three workgroups a CU leave a thread (24 bytes of scratch, and every spilled value
        // behind a wait for ALL loads)
        vae_bn_fetch<MULTI, AUX>(bnr, a.bn_in, a.K, tid);
        if (a.bn_in.stats) vae_bn_table(bnr, a.K, tid, invB, coef);
        __builtin_amdgcn_sched_barrier(0);
    }
    vae_wfetch(w0, 0, tid, wfetch);
    if (nchunks > 1) vae_wfetch(w1, 1, tid, wfetch);
    if (!EARLY) vae_bn_fetch<MULTI, AUX>(bnr, a.bn_in, a.K, tid);
    if (bb0 < rows) zstore(bb0, z0);
    for (int bb = bb0 + 128; bb < rows; bb += 128) { // slices of more than 128 rows
        float4 zv[2];
        zload(bb, zv);
        zstore(bb, zv);
    }
    if (!EARLY && a.bn_in.stats) vae_bn_table(bnr, a.K, tid, invB, coef);
    lrb_barrier();
    {
        // bias gradient of this slice: thread (r, c) sums rows c, c+16, ... of column r
        float sb = 0.0f;
        for (int bb = c; bb < rows; bb += 16) sb += As[r * lda + bb];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) sb += __shfl_xor(sb, o, 64);
        if (c == 0) {
            dbs[r] = sb;
            if (n0 + r < a.N && vw.by < vw.nby) part[a.b_off + n0 + r] = sb;
        }
    }
    v4f_t acc[2];
    for (int ch = 0; ch < nchunks; ++ch) {
        const int k0 = (ch / nR) * VT_N, r0 = (ch % nR) * VT_KC;
        const int rc = rows - r0 < VT_KC ? rows - r0 : VT_KC;
        lrb_barrier();
        if (ch & 1) {
            vae_wstore(w1, ch, tid, Bs, nofix);
            if (ch + 2 < nchunks) vae_wfetch(w1, ch + 2, tid, wfetch);
        } else {
            vae_wstore(w0, ch, tid, Bs, nofix);
            if (ch + 2 < nchunks) vae_wfetch(w0, ch + 2, tid, wfetch);
        }
        lrb_barrier();
        if (r0 == 0) acc[0] = acc[1] = v4f_t{0.0f, 0.0f, 0.0f, 0.0f};
        vae_tile_mfma(As, lda, r0, Bs, rc, lane, wave, acc);
        if (r0 + VT_KC < rows) continue;
        float outv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + vae_ocol(lane, wave, j);
            float v = acc[j >> 2][j & 3];
            if (a.bn_in.stats && k < a.K) v = fmaf(coef[k], v, coef[a.K + k] * dbs[vae_orow(lane, j)]);
            outv[j] = v;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + vae_ocol(lane, wave, j), n = n0 + vae_orow(lane, j);
            if (n < a.N && k < a.K && vw.by < vw.nby) part[a.w_off + (size_t)n * a.K + k] = outv[j];
        }
    }
}

// The NEXT step's batch -- rows perm[pos + B .. pos + 2B) of the data matrix into the other parity's buffer -- is
// fetched by extra workgroups of the dW launch (blockIdx.y >= slices).  It was the tail of the optimiser kernel: a
// chain state -> perm -> row of three dependent round trips, walked B K0 / n_params times by every thread (three times
// at 1024 rows of the 168-column network, seventeen times at 8192: Adam 9.8 and 24 us there against 7.1 for the
// 42-column network).  Nothing in a step reads that buffer, the dW launch is the longest of the step and its
// workgroups are many and short, so the fetch costs nothing here: eight elements per thread, loads issued together.
struct vae_gather_args {
    const float *data;
    const long long *perm;
    float *batch;          // the other parity's batch buffer
    int K0;
};
#define VAE_GATHER_PER_WG 2048

__device__ __forceinline__ void vae_gather_next(const vae_gather_args &g, const vae_state *state, int B, size_t wg, int tid)
{
    const unsigned long long pos = state->pos + (unsigned long long)B, limit = state->limit;
    const size_t total = (size_t)B * g.K0, base = wg * VAE_GATHER_PER_WG + (size_t)tid;
    long long row[8];
    float val[8];
    bool ok[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const size_t i = base + (size_t)u * 256, b = i / (size_t)g.K0;
        ok[u] = i < total && pos + b < limit;
        row[u] = ok[u] ? g.perm[pos + b] : 0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const size_t i = base + (size_t)u * 256, b = i / (size_t)g.K0, k = i - b * (size_t)g.K0;
        val[u] = ok[u] ? g.data[(size_t)row[u] * g.K0 + k] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (ok[u]) g.batch[base + (size_t)u * 256] = val[u];
}

template <bool MULTI, int OCC = 3, bool EARLY = false>   // OCC: workgroups per CU the register budget is held to (two were tried: slower)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void vae_bwd_dw_kernel(const vae_dw_desc *__restrict__ descs, int n_layers, float *part_all,
                                                         size_t n_params, int B, int rows_per_slice, const vae_state *state,
                                                         uint32_t seed, uint32_t keep_threshold, float keep_scale,
                                                         vae_gather_args gather, int slices)
{
    extern __shared__ __attribute__((aligned(16))) float smem_dyn[];
    __shared__ float wsum_st[4][2];
    if ((int)blockIdx.y >= slices) { // (only when gather.data is set: the launch then has the extra rows)
        vae_gather_next(gather, state, B, (size_t)(blockIdx.y - slices) * gridDim.x + blockIdx.x, (int)threadIdx.x);
        return;
    }
    const vae_vwg vw{(int)threadIdx.x, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x, slices, smem_dyn, &wsum_st[0][0]};
    vae_bwd_dw_body<MULTI, false, EARLY>(descs, n_layers, part_all, n_params, B, rows_per_slice, state, seed, keep_threshold, keep_scale, vw);
}

// ---------------------------------------------------------------------------
// optimiser + housekeeping
// ---------------------------------------------------------------------------
struct vae_bn_desc {
    int n;
    unsigned g_off, beta_off; // in the parameter vector
    unsigned run_off;         // running mean at run_off, running var at run_off + n
    unsigned stats_off;       // forward sums at stats_off (2n), backward sums at stats_off + 2n (2n)
};

struct vae_adam_args {
    float *params, *m, *v, *wt, *wp;
    const uint32_t *tpos;    // position of every weight element in the K-major mirror (or ~0)
    const uint32_t *tpos2;   // ... and in the row-padded copy
    const float *part;
    size_t n_params;
    int slices;
    float *running;          // BatchNorm running mean / var
    float *stats;            // all per-step sums (copy 0 of `reps` copies, rep_stride floats apart: vae_bn)
    size_t n_stats;
    unsigned rep_stride;
    const vae_bn_desc *bns;
    int n_bn;
    const vae_state *state;  // this step's counters (this parity)
    vae_state *state_next;   // written for the next step (other parity: nobody reads it during this step)
    float lr, beta1, beta2, eps;
    int B;
    // housekeeping for the next step (see the end of the kernel)
    int K0;                  // width of the data matrix
    const float *data;
    const long long *perm;
    float *batch;
    float *sums_part, *sums;
    int n_wg, n_wg_loss;
    float w_cov, w_comp, w_kld;
};

template <bool MULTI, bool PX = false>
__device__ __forceinline__ void vae_adam_body(vae_adam_args a, const vae_vwg &vw)
{
    constexpr int AUX = PX ? 16 : 0;   // cache policy of the loads (vae_bload4)
    (void)AUX;
    const size_t gid = (size_t)vw.bx * 256 + vw.tid;
    const size_t stride = (size_t)vw.nbx * 256;
    const __amdgpu_buffer_rsrc_t srs = vae_rsrc_reps<MULTI>(a.stats, a.n_stats, a.rep_stride);
    const unsigned long long st_step = vae_ldg<PX>(&a.state->step), st_pos = vae_ldg<PX>(&a.state->pos), st_limit = vae_ldg<PX>(&a.state->limit);
    const unsigned long long t = st_step + 1;
    const float bc1 = 1.0f - powf(a.beta1, (float)t), bc2 = 1.0f - powf(a.beta2, (float)t);
    const float step_size = a.lr / bc1, inv_sqrt_bc2 = 1.0f / sqrtf(bc2);
    // the BatchNorm affine gradients are the backward sums: d(beta) = sum dY, d(gamma) = sum dY xhat
    for (size_t p = gid; p < a.n_params; p += stride) {
        // the element's own state first: these loads depend on nothing and are in flight while the gradient is summed
        const float m0 = a.m[p], v0 = a.v[p], p0 = a.params[p];
        const uint32_t tp = a.tpos[p], tp2 = a.tpos2[p];
        float g = 0.0f;
        bool is_bn = false;
        for (int q = 0; q < a.n_bn; ++q) {
            const vae_bn_desc d = a.bns[q];
            if (p >= d.g_off && p < d.g_off + (unsigned)d.n) {
                g = vae_bsum4<MULTI, AUX>(srs, (uint32_t)(d.stats_off + 2 * d.n + d.n + (p - d.g_off)), a.rep_stride);
                is_bn = true;
            } else if (p >= d.beta_off && p < d.beta_off + (unsigned)d.n) {
                g = vae_bsum4<MULTI, AUX>(srs, (uint32_t)(d.stats_off + 2 * d.n + (p - d.beta_off)), a.rep_stride);
                is_bn = true;
            }
        }
        if (!is_bn) {
            // the slices' partials, eight loads in flight at a time (one load per iteration behind a wait was
            // a chain of `slices` round trips: 8 at batch 1024, 64 at 8192); past the last slice the range
            // check returns zeros, and the sum keeps its order
            // (large batches, 16-64 slices: 32 in flight -- eight at a time was eight round trips at 8192 rows)
            constexpr int FL = MULTI ? 32 : 8;
            for (int s0 = 0; s0 < a.slices; s0 += FL) {
                const int left = a.slices - s0 < FL ? a.slices - s0 : FL;
                const __amdgpu_buffer_rsrc_t prs = vae_rsrc(a.part + (size_t)s0 * a.n_params, (size_t)left * a.n_params);
                float t[FL];
#pragma unroll
                for (int i = 0; i < FL; ++i) t[i] = vae_bload1<AUX>(prs, (uint32_t)((size_t)i * a.n_params + p));
#pragma unroll
                for (int i = 0; i < FL; ++i) g += t[i];
            }
        }
        const float m = a.beta1 * m0 + (1.0f - a.beta1) * g;
        const float v = a.beta2 * v0 + (1.0f - a.beta2) * g * g;
        a.m[p] = m;
        a.v[p] = v;
        const float np_ = p0 - step_size * m / (sqrtf(v) * inv_sqrt_bc2 + a.eps);
        a.params[p] = np_;
        if (tp != 0xFFFFFFFFu) {
            a.wt[tp] = np_;
            a.wp[tp2] = np_;
        }
    }
    // running statistics: momentum 0.1, unbiased variance (torch.nn.BatchNorm1d)
    const float invB = 1.0f / (float)a.B;
    const float unbias = a.B > 1 ? (float)a.B / (float)(a.B - 1) : 1.0f;
    for (int q = 0; q < a.n_bn; ++q) {
        const vae_bn_desc d = a.bns[q];
        for (size_t i = gid; i < (size_t)d.n; i += stride) {
            const float mean = vae_bsum4<MULTI, AUX>(srs, (uint32_t)(d.stats_off + i), a.rep_stride) * invB;
            float var = vae_bsum4<MULTI, AUX>(srs, (uint32_t)(d.stats_off + d.n + i), a.rep_stride) * invB - mean * mean;
            var = var > 0.0f ? var : 0.0f;
            a.running[d.run_off + i] = 0.9f * a.running[d.run_off + i] + 0.1f * mean;
            a.running[d.run_off + d.n + i] = 0.9f * a.running[d.run_off + d.n + i] + 0.1f * var * unbias;
        }
    }
    // ---- housekeeping for the NEXT step.  Everything a step accumulates or counts with lives
    //      twice, by step parity (per-step sums, counters, batch), so nothing written here is
    //      read by this step and no workgroup has to wait for the others: the next batch is
    //      fetched into the other buffer (no kernel of the next step then starts with a
    //      perm -> row dependent load chain), the other parity's sums were cleared by this step's
    //      first kernel, and the counters are written for the other parity.
    {
        const unsigned long long pos = st_pos + (unsigned long long)a.B, limit = st_limit;
        const size_t total = a.data ? (size_t)a.B * a.K0 : 0;   // (data == nullptr: the dW launch has fetched the batch)
        for (size_t i = gid; i < total; i += stride) {
            const size_t b = i / a.K0, k = i - b * a.K0;
            if (pos + b < limit) a.batch[i] = a.data[(size_t)a.perm[pos + b] * a.K0 + k];
        }
        if (gid == 0) {
            a.state_next->step = st_step + 1;
            a.state_next->pos = pos;
            a.state_next->limit = limit;
        }
    }
    if (vw.bx == 0 && vw.tid < 64) { // this step's loss terms into the running totals
        float ec = 0.0f, ep = 0.0f, kl = 0.0f;
        for (int w = vw.tid; w < a.n_wg_loss; w += 64) { // the output layer may have run one workgroup per column chunk
            ec += vae_ldg<PX>(&a.sums_part[w * 4 + 1]);
            ep += vae_ldg<PX>(&a.sums_part[w * 4 + 2]);
        }
        for (int w = vw.tid; w < a.n_wg; w += 64) kl += vae_ldg<PX>(&a.sums_part[w * 4 + 3]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            ec += __shfl_xor(ec, o, 64);
            ep += __shfl_xor(ep, o, 64);
            kl += __shfl_xor(kl, o, 64);
        }
        if (vw.tid == 0) {
            a.sums[0] += a.w_cov * ec + a.w_comp * ep + a.w_kld * kl;
            a.sums[1] += ec;
            a.sums[2] += ep;
            a.sums[3] += kl;
        }
    }
}

template <bool MULTI>
__global__ __launch_bounds__(256) void vae_adam_kernel(vae_adam_args a)
{
    extern __shared__ __attribute__((aligned(16))) float smem_dyn[];
    __shared__ float wsum_st[4][2];
    const vae_vwg vw{(int)threadIdx.x, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x, (int)gridDim.y, smem_dyn, &wsum_st[0][0]};
    vae_adam_body<MULTI, false>(a, vw);
}

__global__ __launch_bounds__(256) void vae_mirror_kernel(const float *params, const uint32_t *tpos, const uint32_t *tpos2,
                                                         float *wt, float *wp, size_t n)
{
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p < n && tpos[p] != 0xFFFFFFFFu) {
        wt[tpos[p]] = params[p];
        wp[tpos2[p]] = params[p];
    }
}

// rows perm[pos .. pos + B) of the data matrix -> batch [B][K]
__global__ __launch_bounds__(256) void vae_gather_kernel(const float *__restrict__ data, const long long *__restrict__ perm,
                                                         const vae_state *state, float *__restrict__ batch, int B, int K)
{
    const unsigned long long pos = state->pos, limit = state->limit;
    const size_t total = (size_t)B * K;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t b = i / K, k = i - b * K;
        if (pos + b < limit) batch[i] = data[(size_t)perm[pos + b] * K + k];
    }
}

// ---------------------------------------------------------------------------
// host: the trainer object
// ---------------------------------------------------------------------------
struct vae_dense {
    int K, N;
    size_t w_off, b_off;   // in the parameter vector: W[N][K], b[N]
    size_t wt_off, wp_off; // internal copies: transposed [K][N4] and row-padded [N][K4], 16-byte aligned rows
};

struct lrb_vae {
    lrb_ctx *ctx;
    int d0, cov_size, latent, n_hidden;
    std::vector<int> hidden;
    std::vector<vae_dense> enc, dec; // blocks
    vae_dense heads, outl;
    std::vector<vae_bn_desc> bns;    // enc blocks then dec blocks
    size_t n_params, n_running, n_stats;
    bool no_fuse, no_narrow, fuse_dz0;
    bool det;          // LRB_VAE_DETERMINISTIC=1: the batch sums added up in a fixed order (vae_stat_add, vae_det_reduce_kernel)
    float *det_part;   // [row tiles of the largest batch][n_stats]
    int max_batch, max_slices;
    float w_cov, w_comp, w_kld, lr, dropout;
    uint32_t seed;
    // device memory
    float *params, *m, *v, *running, *stats, *sums, *part, *wt, *wp;
    uint32_t *d_tpos, *d_tpos2;
    size_t n_wt, n_wp;
    vae_dw_desc *d_dw;
    int n_dw, dw_tiles, dw_kmax;
    const float *graph_data;
    const int64_t *graph_perm;
    vae_bn_desc *d_bns;
    vae_state *state;
    std::vector<float *> act_enc, act_dec, dY_enc, dY_dec, dZ_enc, dZ_dec;
    float *heads_out, *z, *eps, *dz, *dheads, *grad_out, *batch, *sums_part, *eval_stats;
    unsigned long long host_steps; // steps enqueued so far: its parity selects the buffers of the next step
    // graphs per (batch size, step parity)
    std::vector<int> graph_B;
    std::vector<hipGraphExec_t> graph_exec;
    hipStream_t cap_stream;
};

// LDS of the forward / dX kernels: tile [16][w+1], B chunk, reduction scratch, coefficient tables (up to 5 per column
// of either width)
static size_t vae_fwd_smem(int w, int w2)
{
    const size_t w4 = ((size_t)w + 3) & ~(size_t)3;
    return ((size_t)((VT_M * (w4 + 1) + 3) & ~3) + VT_KC * VT_NS + 5 * (size_t)w + 2 * (size_t)w2) * 4;
}

template <typename T> static int vae_alloc(T **p, size_t count)
{
    HIP_TRY(hipMalloc((void **)p, count * sizeof(T) + 64));
    HIP_TRY(hipMemset(*p, 0, count * sizeof(T) + 64));
    return LRB_OK;
}

extern "C" int lrb_vae_destroy(lrb_vae *v)
{
    if (!v) return LRB_OK;
    for (hipGraphExec_t g : v->graph_exec) (void)hipGraphExecDestroy(g);
    if (v->cap_stream) (void)hipStreamDestroy(v->cap_stream);
    void *single[] = {v->params, v->m, v->v, v->running, v->stats, v->sums, v->part, v->wt, v->wp, v->d_tpos, v->d_tpos2, v->d_dw,
                      v->d_bns, v->state, v->det_part,
                      v->heads_out, v->z, v->eps, v->dz, v->dheads, v->grad_out, v->batch, v->sums_part, v->eval_stats};
    for (void *p : single)
        if (p) (void)hipFree(p);
    for (auto *vec : {&v->act_enc, &v->act_dec, &v->dY_enc, &v->dY_dec, &v->dZ_enc, &v->dZ_dec})
        for (float *p : *vec)
            if (p) (void)hipFree(p);
    delete v;
    return LRB_OK;
}

extern "C" int lrb_vae_create(lrb_ctx *c, int cov_size, int prof_size, const int *hidden, int n_hidden, int latent,
                              int max_batch, const float *loss_weights, float lr, float dropout, uint64_t seed,
                              lrb_vae **out)
{
    ARG_TRY(c != nullptr && out != nullptr && hidden != nullptr && loss_weights != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    ARG_TRY(cov_size >= 0 && prof_size >= 0 && cov_size + prof_size >= 1 && cov_size + prof_size <= VAE_MAX_WIDTH);
    ARG_TRY(n_hidden >= 1 && n_hidden <= 6 && latent >= 1 && latent <= 256);
    ARG_TRY(max_batch >= 2 && max_batch <= (1 << 20));
    ARG_TRY(dropout >= 0.0f && dropout < 1.0f && lr > 0.0f);
    for (int i = 0; i < n_hidden; ++i) ARG_TRY(hidden[i] >= 1 && hidden[i] <= VAE_MAX_WIDTH);
    {
        // the kernels address every [batch][width] matrix through a 32-bit buffer offset
        uint64_t widest = (uint64_t)(cov_size + prof_size);
        for (int i = 0; i < n_hidden; ++i) widest = hidden[i] > (int)widest ? (uint64_t)hidden[i] : widest;
        widest = (uint64_t)(2 * latent) > widest ? (uint64_t)(2 * latent) : widest;
        ARG_TRY((uint64_t)max_batch * widest * 4 < (1ull << 31));
    }
    HIP_TRY(hipSetDevice(c->device));
    lrb_vae *v = new lrb_vae();
    v->ctx = c;
    v->d0 = cov_size + prof_size;
    v->cov_size = cov_size;
    v->latent = latent;
    v->no_fuse = getenv("LRB_VAE_NO_FUSE") && atoi(getenv("LRB_VAE_NO_FUSE")); // debugging: one launch per layer
    v->no_narrow = false;          // (128-column tiles only: round 3's A/B, 64-column tiles won where a layer is narrow)
    v->fuse_dz0 = !v->no_fuse;
    // repeatable runs: no float atomics -- the same seed gives the same parameters, bit for bit, run after run (slower:
    // one more small launch behind every kernel that adds to the batch sums)
    v->det = getenv("LRB_VAE_DETERMINISTIC") && atoi(getenv("LRB_VAE_DETERMINISTIC"));
    v->det_part = nullptr;
    v->n_hidden = n_hidden;
    v->hidden.assign(hidden, hidden + n_hidden);
    v->max_batch = max_batch;
    v->max_slices = (max_batch + 127) / 128;
    v->w_cov = loss_weights[0];
    v->w_comp = loss_weights[1];
    v->w_kld = loss_weights[2];
    v->lr = lr;
    v->dropout = dropout;
    v->seed = (uint32_t)(seed ^ (seed >> 32));
    v->cap_stream = nullptr;
    v->params = v->m = v->v = v->running = v->stats = v->sums = v->part = v->wt = v->wp = nullptr;
    v->d_tpos = v->d_tpos2 = nullptr;
    v->d_dw = nullptr;
    v->host_steps = 0;
    v->graph_data = nullptr;
    v->graph_perm = nullptr;
    v->d_bns = nullptr;
    v->state = nullptr;
    v->heads_out = v->z = v->eps = v->dz = v->dheads = v->grad_out = v->batch = v->sums_part = v->eval_stats = nullptr;
    // parameter vector: per block W, b, gamma, beta; heads [Wmu; Wls], [bmu; bls]; ...; output W, b
    size_t off = 0, run = 0, st = 0;
    auto add_block = [&](std::vector<vae_dense> &list, int K, int N) {
        vae_dense d{K, N, off, off + (size_t)N * K, 0, 0};
        off += (size_t)N * K + N;
        vae_bn_desc b{N, (unsigned)off, (unsigned)(off + N), (unsigned)run, (unsigned)st};
        off += 2 * (size_t)N;
        run += 2 * (size_t)N;
        st += 4 * (size_t)N;
        list.push_back(d);
        v->bns.push_back(b);
    };
    int K = v->d0;
    for (int i = 0; i < n_hidden; ++i) {
        add_block(v->enc, K, hidden[i]);
        K = hidden[i];
    }
    v->heads = vae_dense{K, 2 * latent, off, off + (size_t)2 * latent * K, 0, 0};
    off += (size_t)2 * latent * K + 2 * latent;
    K = latent;
    for (int i = n_hidden - 1; i >= 0; --i) {
        add_block(v->dec, K, hidden[i]);
        K = hidden[i];
    }
    v->outl = vae_dense{K, v->d0, off, off + (size_t)v->d0 * K, 0, 0};
    off += (size_t)v->d0 * K + v->d0;
    v->n_params = off;
    v->n_running = run;
    v->n_stats = st;
    {
        size_t to = 0, po = 0;
        auto place = [&](vae_dense &L) {
            L.wt_off = to;
            to += (size_t)L.K * ((L.N + 3) & ~3);
            L.wp_off = po;
            po += (size_t)L.N * ((L.K + 3) & ~3);
        };
        for (vae_dense &L : v->enc) place(L);
        for (vae_dense &L : v->dec) place(L);
        place(v->heads);
        place(v->outl);
        v->n_wt = to;
        v->n_wp = po;
    }
    int rc = LRB_OK;
    auto A = [&](float **p, size_t n) {
        if (rc == LRB_OK) rc = vae_alloc(p, n);
    };
    A(&v->params, v->n_params);
    A(&v->wt, v->n_wt);
    A(&v->wp, v->n_wp);
    A(&v->m, v->n_params);
    A(&v->v, v->n_params);
    A(&v->running, v->n_running);
    A(&v->stats, 2 * VAE_REPS * v->n_stats); // per-step sums: VAE_REPS copies per step parity
    if (v->det) A(&v->det_part, (size_t)((max_batch + VT_M - 1) / VT_M) * v->n_stats);
    A(&v->sums, 4);
    A(&v->part, (size_t)v->max_slices * v->n_params);
    const size_t Bm = (size_t)max_batch;
    for (int i = 0; i < n_hidden; ++i) {
        float *p = nullptr;
        A(&p, Bm * v->enc[i].N); v->act_enc.push_back(p); p = nullptr;
        A(&p, Bm * v->enc[i].N); v->dY_enc.push_back(p); p = nullptr;
        A(&p, Bm * v->enc[i].N); v->dZ_enc.push_back(p); p = nullptr;
        A(&p, Bm * v->dec[i].N); v->act_dec.push_back(p); p = nullptr;
        A(&p, Bm * v->dec[i].N); v->dY_dec.push_back(p); p = nullptr;
        A(&p, Bm * v->dec[i].N); v->dZ_dec.push_back(p);
    }
    A(&v->heads_out, Bm * 2 * latent);
    A(&v->z, Bm * latent);
    A(&v->eps, Bm * latent);
    A(&v->dz, Bm * latent);
    A(&v->dheads, Bm * 2 * latent);
    A(&v->grad_out, Bm * v->d0);
    A(&v->batch, 2 * Bm * v->d0);       // the gathered batch, one per step parity
    A(&v->sums_part, ((Bm + VT_M - 1) / VT_M) * 4 * ((VAE_MAX_WIDTH + 63) / 64));
    A(&v->eval_stats, v->n_stats);
    if (rc == LRB_OK && hipMalloc((void **)&v->d_bns, v->bns.size() * sizeof(vae_bn_desc)) != hipSuccess) rc = LRB_ERR_NOMEM;
    if (rc == LRB_OK && hipMalloc((void **)&v->state, 2 * sizeof(vae_state)) != hipSuccess) rc = LRB_ERR_NOMEM;
    if (rc == LRB_OK && hipMalloc((void **)&v->d_tpos, v->n_params * sizeof(uint32_t)) != hipSuccess) rc = LRB_ERR_NOMEM;
    if (rc == LRB_OK && hipMalloc((void **)&v->d_tpos2, v->n_params * sizeof(uint32_t)) != hipSuccess) rc = LRB_ERR_NOMEM;
    if (rc == LRB_OK) {
        std::vector<uint32_t> tpos(v->n_params, 0xFFFFFFFFu), tpos2(v->n_params, 0xFFFFFFFFu);
        auto mirror = [&](const vae_dense &L) {
            const size_t N4 = (L.N + 3) & ~3, K4 = (L.K + 3) & ~3;
            for (int n = 0; n < L.N; ++n)
                for (int k = 0; k < L.K; ++k) {
                    tpos[L.w_off + (size_t)n * L.K + k] = (uint32_t)(L.wt_off + (size_t)k * N4 + n);
                    tpos2[L.w_off + (size_t)n * L.K + k] = (uint32_t)(L.wp_off + (size_t)n * K4 + k);
                }
        };
        for (const vae_dense &L : v->enc) mirror(L);
        for (const vae_dense &L : v->dec) mirror(L);
        mirror(v->heads);
        mirror(v->outl);
        (void)hipMemcpy(v->d_tpos, tpos.data(), tpos.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(v->d_tpos2, tpos2.data(), tpos2.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(v->d_bns, v->bns.data(), v->bns.size() * sizeof(vae_bn_desc), hipMemcpyHostToDevice);
        (void)hipMemset(v->state, 0, 2 * sizeof(vae_state));
        // running variance starts at 1
        std::vector<float> r(v->n_running, 0.0f);
        for (const vae_bn_desc &b : v->bns)
            for (int i = 0; i < b.n; ++i) r[b.run_off + b.n + i] = 1.0f;
        (void)hipMemcpy(v->running, r.data(), r.size() * 4, hipMemcpyHostToDevice);
        if (hipStreamCreateWithFlags(&v->cap_stream, hipStreamNonBlocking) != hipSuccess) rc = LRB_ERR_HIP;
    }
    if (rc != LRB_OK) {
        lrb_vae_destroy(v);
        if (rc == LRB_ERR_NOMEM) lrb_set_error("out of device memory for the VAE trainer%s%s", "", "");
        return rc;
    }
    {
        // one descriptor per Linear for the batched dW launch, one table per step parity
        std::vector<vae_dw_desc> dd;
        int tile = 0, kmax = 1;
        const int nh = v->n_hidden;
        for (int par = 0; par < 2; ++par) {
            tile = 0;
            float *stats = v->stats + (size_t)par * VAE_REPS * v->n_stats;
            const float *batch = v->batch + (size_t)par * max_batch * v->d0;
            auto bn_of = [&](int q) {
                const vae_bn_desc &d = v->bns[q];
                return vae_bn{stats + d.stats_off, v->params + d.g_off, v->params + d.beta_off, (unsigned)v->n_stats};
            };
            const vae_bn none{nullptr, nullptr, nullptr, 0u};
            auto add = [&](const vae_dense &L, const float *dZ, const float *in, vae_bn bn_in) {
                dd.push_back(vae_dw_desc{dZ, in, bn_in, L.w_off, L.b_off, L.K, L.N, tile, nullptr, nullptr, none, nullptr, 0});
                tile += (L.N + VT_M - 1) / VT_M;
                if (L.K > kmax) kmax = L.K;
            };
            add(v->outl, v->grad_out, v->act_dec[nh - 1], bn_of(2 * nh - 1));
            for (int i = nh - 1; i >= 0; --i)
                add(v->dec[i], v->dZ_dec[i], i > 0 ? v->act_dec[i - 1] : v->z, i > 0 ? bn_of(nh + i - 1) : none);
            add(v->heads, v->dheads, v->act_enc[nh - 1], bn_of(nh - 1));
            for (int i = nh - 1; i >= 0; --i)
                add(v->enc[i], v->dZ_enc[i], i > 0 ? v->act_enc[i - 1] : batch, i > 0 ? bn_of(i - 1) : none);
            if (v->fuse_dz0) { // the first encoder block's dZ is made inside the dW launch (no launch of its own)
                vae_dw_desc &e0 = dd.back();
                e0.dY = v->dY_enc[0];
                e0.act = v->act_enc[0];
                e0.bn = bn_of(0);
                e0.bsum = stats + v->bns[0].stats_off + 2 * v->bns[0].n;
                e0.layer = 0;
            }
        }
        v->n_dw = (int)dd.size() / 2;
        v->dw_tiles = tile;
        v->dw_kmax = kmax;
        HIP_TRY(hipMalloc((void **)&v->d_dw, dd.size() * sizeof(vae_dw_desc)));
        HIP_TRY(hipMemcpy(v->d_dw, dd.data(), dd.size() * sizeof(vae_dw_desc), hipMemcpyHostToDevice));
    }
    // kernels whose LDS tile exceeds the default limit
    const size_t big = vae_fwd_smem(VAE_MAX_WIDTH, VAE_MAX_WIDTH);
#define VAE_BIG_SMEM(k) HIP_TRY(hipFuncSetAttribute((const void *)(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)big))
    VAE_BIG_SMEM((vae_fwd_kernel<VAE_ACT_BLOCK, false, 2>)); VAE_BIG_SMEM((vae_fwd_kernel<VAE_ACT_BLOCK, true, 2>));
    VAE_BIG_SMEM((vae_fwd_kernel<VAE_ACT_HEADS, false, 2>)); VAE_BIG_SMEM((vae_fwd_kernel<VAE_ACT_HEADS, true, 2>));
    VAE_BIG_SMEM((vae_fwd_kernel<VAE_ACT_LOSS, false, 2>)); VAE_BIG_SMEM((vae_fwd_kernel<VAE_ACT_LOSS, true, 2>));
    VAE_BIG_SMEM((vae_fwd_kernel<VAE_ACT_BLOCK, false, 1>)); VAE_BIG_SMEM((vae_fwd_kernel<VAE_ACT_BLOCK, true, 1>));
    VAE_BIG_SMEM((vae_fwd_kernel<VAE_ACT_HEADS, false, 1>)); VAE_BIG_SMEM((vae_fwd_kernel<VAE_ACT_HEADS, true, 1>));
    VAE_BIG_SMEM((vae_fwd_kernel<VAE_ACT_LOSS, false, 1>)); VAE_BIG_SMEM((vae_fwd_kernel<VAE_ACT_LOSS, true, 1>));
    VAE_BIG_SMEM((vae_bwd_dx_kernel<false, false, 2>)); VAE_BIG_SMEM((vae_bwd_dx_kernel<false, true, 2>));
    VAE_BIG_SMEM((vae_bwd_dx_kernel<true, false, 2>)); VAE_BIG_SMEM((vae_bwd_dx_kernel<true, true, 2>));
    VAE_BIG_SMEM((vae_bwd_dx_kernel<false, false, 1>)); VAE_BIG_SMEM((vae_bwd_dx_kernel<false, true, 1>));
    VAE_BIG_SMEM((vae_bwd_dx_kernel<true, false, 1>)); VAE_BIG_SMEM((vae_bwd_dx_kernel<true, true, 1>));
    HIP_TRY(hipFuncSetAttribute((const void *)vae_bwd_dw_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)VAE_DW_SMEM_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)vae_bwd_dw_kernel<true, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)VAE_DW_SMEM_MAX));
#undef VAE_BIG_SMEM
    *out = v;
    return LRB_OK;
}

extern "C" int lrb_vae_sizes(lrb_vae *v, uint64_t *n_params, uint64_t *n_running)
{
    ARG_TRY(v != nullptr);
    if (n_params) *n_params = v->n_params;
    if (n_running) *n_running = v->n_running;
    return LRB_OK;
}

// what: 0 parameters, 1 BatchNorm running statistics, 2 Adam m, 3 Adam v, 4 loss sums (4 floats)
static int vae_buf(lrb_vae *v, int what, float **p, size_t *n)
{
    switch (what) {
    case 0: *p = v->params; *n = v->n_params; return LRB_OK;
    case 1: *p = v->running; *n = v->n_running; return LRB_OK;
    case 2: *p = v->m; *n = v->n_params; return LRB_OK;
    case 3: *p = v->v; *n = v->n_params; return LRB_OK;
    case 4: *p = v->sums; *n = 4; return LRB_OK;
    default:
        lrb_set_error("invalid argument: unknown VAE buffer%s%s", "", "");
        return LRB_ERR_ARG;
    }
}

extern "C" int lrb_vae_set(lrb_vae *v, int what, const float *host, uint64_t count)
{
    ARG_TRY(v != nullptr && host != nullptr);
    float *p;
    size_t n;
    int rc = vae_buf(v, what, &p, &n);
    if (rc != LRB_OK) return rc;
    ARG_TRY(count == n);
    HIP_TRY(hipStreamSynchronize(v->ctx->stream));
    HIP_TRY(hipMemcpy(p, host, n * 4, hipMemcpyHostToDevice));
    if (what == 0) {
        hipLaunchKernelGGL(vae_mirror_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, v->ctx->stream, v->params,
                           v->d_tpos, v->d_tpos2, v->wt, v->wp, n);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(v->ctx->stream));
    }
    return LRB_OK;
}

extern "C" int lrb_vae_get(lrb_vae *v, int what, float *host, uint64_t count)
{
    ARG_TRY(v != nullptr && host != nullptr);
    float *p;
    size_t n;
    int rc = vae_buf(v, what, &p, &n);
    if (rc != LRB_OK) return rc;
    ARG_TRY(count == n);
    HIP_TRY(hipStreamSynchronize(v->ctx->stream));
    HIP_TRY(hipMemcpy(host, p, n * 4, hipMemcpyDeviceToHost));
    return LRB_OK;
}

extern "C" int lrb_vae_steps_done(lrb_vae *v, uint64_t *steps)
{
    ARG_TRY(v != nullptr && steps != nullptr);
    HIP_TRY(hipStreamSynchronize(v->ctx->stream));
    *steps = v->host_steps;
    return LRB_OK;
}

static bool g_vae_sync_each = false;
#define VAE_DBG_SYNC()                                                  \
    do {                                                                \
        if (g_vae_sync_each) HIP_TRY(hipDeviceSynchronize());           \
    } while (0)

// LDS of one dW workgroup: the dZ^T tile of its slice, one chunk of activations, the BatchNorm table, the bias sums
static size_t vae_dw_smem(int rows, int kmax)
{
    return ((size_t)((VT_M * (rows + 1) + 3) & ~3) + VT_KC * VT_NS + 2 * (size_t)kmax + VT_M) * 4;
}

// How the batch is cut into slices for the dW launch (one partial gradient per slice, summed by the optimiser): 128
// rows a slice.  Fewer, longer slices were tried for the large batches (16-64 partials of 316-357 KB each are written
// and read back there): the step does not get shorter -- 8192 rows, C1 shape: 64 slices 176 us, 32: 173, 16: 179, 8:
// 209, 7 (one round of workgroups): 167-182; C3 shape: 221, 219, 224, 264, 280 (profiles/r03_vae_dw_slices.txt) -- a
// workgroup's serial walk over a long slice costs more than the partials it saves.
static void vae_dw_geometry(const lrb_vae *v, int B, int *rows_out, int *slices_out)
{
    auto rows_for = [&](int s) { int r = ((B + s - 1) / s + 3) & ~3; return r < 128 ? 128 : r; };
    int slices = (B + 127) / 128;
    while (slices < (B + 127) / 128 && vae_dw_smem(rows_for(slices), v->dw_kmax) > VAE_DW_SMEM_MAX) ++slices;
    const int rows = rows_for(slices);
    *rows_out = rows;
    *slices_out = (B + rows - 1) / rows;
}

static int vae_enqueue_step(lrb_vae *v, const float *d_data, const long long *d_perm, int B, hipStream_t st, int par)
{
    // everything a step accumulates or counts with exists once per step parity
    float *const stats = v->stats + (size_t)par * VAE_REPS * v->n_stats;
    float *const stats_next = v->stats + (size_t)(par ^ 1) * VAE_REPS * v->n_stats;
    vae_state *const state = v->state + par, *const state_next = v->state + (par ^ 1);
    float *const batch = v->batch + (size_t)par * v->max_batch * v->d0;
    float *const batch_next = v->batch + (size_t)(par ^ 1) * v->max_batch * v->d0;
    // the next step's batch: fetched by extra workgroups of the dW launch (round 2 had the optimiser kernel fetch it)
    const bool gather_in_dw = true;
    const int nh = v->n_hidden;
    const dim3 blk(256), grid((B + VT_M - 1) / VT_M);
    const uint32_t keep_thr = (uint32_t)((double)v->dropout * 4294967296.0);
    const float keep_scale = 1.0f / (1.0f - v->dropout);
    auto bn_of = [&](int q) {
        const vae_bn_desc &d = v->bns[q];
        return vae_bn{stats + d.stats_off, v->params + d.g_off, v->params + d.beta_off, (unsigned)v->n_stats};
    };
    const vae_bn none{nullptr, nullptr, nullptr, 0u};
    // the reduction over the latent dimensions is short: both Linears next to z run inside their neighbours' kernels
    const bool fuse_latent = v->latent <= 64 && !v->no_fuse;
    // from 2048 rows on the float atomics of the batch statistics are spread over copies (vae_bn)
    const bool multi = vae_reps_for(B) > 1 && !v->det;
    const int det = v->det ? 1 : 0;
    const long long det_shift = v->det ? (long long)(v->det_part - stats) : 0;
    // deterministic mode: behind every launch that adds to the batch sums, the row tiles' shares added up in order
    auto det_reduce = [&]() {
        if (det)
            hipLaunchKernelGGL(vae_det_reduce_kernel, dim3((unsigned)((v->n_stats + 255) / 256)), dim3(256), 0, st, (const float *)v->det_part,
                               (int)grid.x, v->n_stats, stats);
    };
    // Column chunks of a layer: 64 columns per workgroup (one MFMA tile per wave) when the layer is that narrow, or
    // when one workgroup per 64-column chunk still leaves CUs idle -- half the weights to stage and half the MFMAs on
    // each kernel's critical path; otherwise 128 columns, one workgroup per chunk while THAT leaves CUs idle, else
    // one workgroup looping over the chunks.
    const bool narrow_ok = !v->no_narrow;
    // (every workgroup of a split builds the whole 16 x red input tile: with a wide reduction -- the first layer at
    //  k = 5 -- the split only pays while it does not put two workgroups on a CU)
    auto col_nt = [&](int N, int red) {
        const unsigned c64 = (unsigned)((N + 63) / 64), cus = (unsigned)v->ctx->n_cu;
        return (narrow_ok && (c64 == 1 || grid.x * c64 <= (red <= 256 ? 2u : 1u) * cus)) ? 1 : 2;
    };
    auto col_grid = [&](int N, int red) {
        const unsigned chunks = (unsigned)((N + 64 * col_nt(N, red) - 1) / (64 * col_nt(N, red)));
        return (chunks > 1 && grid.x * chunks <= 2u * (unsigned)v->ctx->n_cu) ? dim3(grid.x, chunks) : grid;
    };
#define VAE_FWD_LAUNCH(ACT, N_, GRID)                                                                                   \
    do {                                                                                                                \
        a.det = det;                                                                                                    \
        a.det_shift = det_shift;                                                                                        \
        if (col_nt(N_, a.K) == 1) {                                                                                            \
            if (multi) hipLaunchKernelGGL((vae_fwd_kernel<ACT, true, 1>), GRID, blk, vae_fwd_smem(a.K, 0), st, a);      \
            else hipLaunchKernelGGL((vae_fwd_kernel<ACT, false, 1>), GRID, blk, vae_fwd_smem(a.K, 0), st, a);           \
        } else {                                                                                                        \
            if (multi) hipLaunchKernelGGL((vae_fwd_kernel<ACT, true, 2>), GRID, blk, vae_fwd_smem(a.K, 0), st, a);      \
            else hipLaunchKernelGGL((vae_fwd_kernel<ACT, false, 2>), GRID, blk, vae_fwd_smem(a.K, 0), st, a);           \
        }                                                                                                               \
        if (ACT == VAE_ACT_BLOCK) det_reduce();                                                                         \
    } while (0)
    // ---- forward ----
    for (int i = 0; i < nh; ++i) {
        vae_fwd_args a{};
        a.in = i == 0 ? batch : v->act_enc[i - 1]; // the batch was gathered by the previous step's housekeeping
        if (i == 0) {
            a.zero = stats_next;
            a.zero_n = (int)(VAE_REPS * v->n_stats);
        }
        a.bn_in = i == 0 ? none : bn_of(i - 1);
        a.Wt = v->wt + v->enc[i].wt_off;
        a.bias = v->params + v->enc[i].b_off;
        a.out = v->act_enc[i];
        a.stats_out = stats + v->bns[i].stats_off; a.rep_stride = (unsigned)v->n_stats;
        a.state = state;
        a.B = B; a.K = v->enc[i].K; a.N = v->enc[i].N; a.layer = i;
        a.seed = v->seed; a.keep_threshold = keep_thr; a.keep_scale = keep_scale;
        VAE_FWD_LAUNCH(VAE_ACT_BLOCK, a.N, col_grid(a.N, a.K));
    }
    {
        vae_fwd_args a{};
        a.in = v->act_enc[nh - 1];
        a.bn_in = bn_of(nh - 1);
        a.Wt = v->wt + v->heads.wt_off;
        a.bias = v->params + v->heads.b_off;
        a.out = v->heads_out;
        a.z = v->z; a.eps = v->eps; a.sums_part = v->sums_part;
        a.state = state;
        a.B = B; a.K = v->heads.K; a.N = v->heads.N; a.layer = 100;
        a.seed = v->seed;
        if (fuse_latent) { // the first decoder block rides along
            a.nx_Wt = v->wt + v->dec[0].wt_off;
            a.nx_bias = v->params + v->dec[0].b_off;
            a.nx_out = v->act_dec[0];
            a.nx_stats = stats + v->bns[nh].stats_off; a.rep_stride = (unsigned)v->n_stats;
            a.nx_N = v->dec[0].N; a.nx_layer = 50;
            a.keep_threshold = keep_thr; a.keep_scale = keep_scale;
        }
        // one workgroup per row tile (the epilogue needs whole rows): the narrow tile only if the layer is one chunk
        a.det = det;
        a.det_shift = det_shift;
        if (narrow_ok && a.N <= 64) {
            if (multi) hipLaunchKernelGGL((vae_fwd_kernel<VAE_ACT_HEADS, true, 1>), grid, blk, vae_fwd_smem(a.K, 0), st, a);
            else hipLaunchKernelGGL((vae_fwd_kernel<VAE_ACT_HEADS, false, 1>), grid, blk, vae_fwd_smem(a.K, 0), st, a);
        } else {
            if (multi) hipLaunchKernelGGL((vae_fwd_kernel<VAE_ACT_HEADS, true, 2>), grid, blk, vae_fwd_smem(a.K, 0), st, a);
            else hipLaunchKernelGGL((vae_fwd_kernel<VAE_ACT_HEADS, false, 2>), grid, blk, vae_fwd_smem(a.K, 0), st, a);
        }
        det_reduce();   // (the first decoder block's sums, when it runs inside this launch)
    }
    for (int i = fuse_latent ? 1 : 0; i < nh; ++i) {
        vae_fwd_args a{};
        a.in = i == 0 ? v->z : v->act_dec[i - 1];
        a.bn_in = i == 0 ? none : bn_of(nh + i - 1);
        a.Wt = v->wt + v->dec[i].wt_off;
        a.bias = v->params + v->dec[i].b_off;
        a.out = v->act_dec[i];
        a.stats_out = stats + v->bns[nh + i].stats_off; a.rep_stride = (unsigned)v->n_stats;
        a.state = state;
        a.B = B; a.K = v->dec[i].K; a.N = v->dec[i].N; a.layer = 50 + i;
        a.seed = v->seed; a.keep_threshold = keep_thr; a.keep_scale = keep_scale;
        VAE_FWD_LAUNCH(VAE_ACT_BLOCK, a.N, col_grid(a.N, a.K));
    }
    {
        vae_fwd_args a{};
        a.in = v->act_dec[nh - 1];
        a.bn_in = bn_of(2 * nh - 1);
        a.Wt = v->wt + v->outl.wt_off;
        a.bias = v->params + v->outl.b_off;
        a.data = batch;
        a.grad = v->grad_out;
        a.sums_part = v->sums_part;
        a.cov_size = v->cov_size;
        a.w_cov = v->w_cov; a.w_comp = v->w_comp;
        a.state = state;
        a.B = B; a.K = v->outl.K; a.N = v->outl.N; a.layer = 200;
        a.seed = v->seed;
        VAE_FWD_LAUNCH(VAE_ACT_LOSS, a.N, col_grid(a.N, a.K));
    }
#undef VAE_FWD_LAUNCH
    // ---- backward: the dX chain, then every layer's dW in one launch ----
    int rows, slices;
    vae_dw_geometry(v, B, &rows, &slices);
    auto dx = [&](const vae_dense &L, const float *dY, int block_q /* -1: plain layer */, const float *act, float *dZ,
                  float *dX, int below_q /* -1: none */, const float *act_below, int layer, bool latent = false) {
        vae_bwd_args a{};
        if (latent) {
            a.heads = v->heads_out; a.eps = v->eps; a.dheads = v->dheads; a.w_kld = v->w_kld;
            if (fuse_latent) {
                a.h_W = v->wp + v->heads.wp_off;
                a.h_dX = v->dY_enc[nh - 1];
                a.h_act_below = v->act_enc[nh - 1];
                a.h_bn_below = bn_of(nh - 1);
                a.h_bsum_below = stats + v->bns[nh - 1].stats_off + 2 * v->bns[nh - 1].n;
                a.h_K = v->heads.K;
            }
        }
        a.dY = dY; a.act = act; a.dZ = dZ; a.W = v->wp + L.wp_off; a.dX = dX;
        a.block = block_q >= 0;
        if (block_q >= 0) {
            a.bn = bn_of(block_q);
            a.bsum = stats + v->bns[block_q].stats_off + 2 * v->bns[block_q].n;
        }
        if (below_q >= 0) {
            a.act_below = act_below;
            a.bn_below = bn_of(below_q);
            a.bsum_below = stats + v->bns[below_q].stats_off + 2 * v->bns[below_q].n;
        }
        a.state = state;
        a.B = B; a.K = L.K; a.N = L.N; a.layer = layer;
        a.seed = v->seed; a.keep_threshold = keep_thr; a.keep_scale = keep_scale; a.rep_stride = (unsigned)v->n_stats;
        // the output columns of dX are the layer's K inputs: the same choice of column chunks as the forward kernels
        // (a launch without dX only builds the dZ tile: one workgroup per row tile)
        const int nt = dX ? col_nt(L.K, L.N) : 2;
        const dim3 dgrid = (dX && !latent) ? col_grid(L.K, L.N) : grid;
#define VAE_DX_LAUNCH(LAT, MUL)                                                                                        \
    do {                                                                                                                \
        if (nt == 1) hipLaunchKernelGGL((vae_bwd_dx_kernel<LAT, MUL, 1>), dgrid, blk, vae_fwd_smem(L.N, L.K), st, a);   \
        else hipLaunchKernelGGL((vae_bwd_dx_kernel<LAT, MUL, 2>), dgrid, blk, vae_fwd_smem(L.N, L.K), st, a);           \
    } while (0)
        a.det = det;
        a.det_shift = det_shift;
        if (latent && multi) VAE_DX_LAUNCH(true, true);
        else if (latent) VAE_DX_LAUNCH(true, false);
        else if (multi) VAE_DX_LAUNCH(false, true);
        else VAE_DX_LAUNCH(false, false);
#undef VAE_DX_LAUNCH
        det_reduce();
        if (g_vae_sync_each) (void)hipDeviceSynchronize();
    };
    // output layer: dZ = dL/drecon
    dx(v->outl, v->grad_out, -1, nullptr, nullptr, v->dY_dec[nh - 1], 2 * nh - 1, v->act_dec[nh - 1], 200);
    for (int i = nh - 1; i >= 0; --i) {
        float *dX = i > 0 ? v->dY_dec[i - 1] : v->dz;
        dx(v->dec[i], v->dY_dec[i], nh + i, v->act_dec[i], v->dZ_dec[i], dX, i > 0 ? nh + i - 1 : -1,
           i > 0 ? v->act_dec[i - 1] : nullptr, 50 + i, i == 0);
    }
    if (!fuse_latent) dx(v->heads, v->dheads, -1, nullptr, nullptr, v->dY_enc[nh - 1], nh - 1, v->act_enc[nh - 1], 100);
    for (int i = nh - 1; i >= (v->fuse_dz0 ? 1 : 0); --i)   // (the first block's dZ: inside the dW launch when fused)
        dx(v->enc[i], v->dY_enc[i], i, v->act_enc[i], v->dZ_enc[i], i > 0 ? v->dY_enc[i - 1] : nullptr, i > 0 ? i - 1 : -1,
           i > 0 ? v->act_enc[i - 1] : nullptr, i);
    {
        const size_t smem = vae_dw_smem(rows, v->dw_kmax);
        {
            // + the rows of workgroups that fetch the next step's batch (vae_gather_next)
            const size_t per_row = (size_t)v->dw_tiles * VAE_GATHER_PER_WG;
            const int grows = gather_in_dw ? (int)(((size_t)B * v->d0 + per_row - 1) / per_row) : 0;
            const vae_gather_args ga{gather_in_dw ? d_data : nullptr, d_perm, batch_next, v->d0};
            if (multi)   // (large batches: the BatchNorm table first -- no spill, 3 % faster at 8192 rows: profiles/r06_vae_dw_ab.txt)
                hipLaunchKernelGGL((vae_bwd_dw_kernel<true, 3, true>), dim3(v->dw_tiles, slices + grows), blk, smem, st, v->d_dw + (size_t)par * v->n_dw, v->n_dw,
                                   v->part, v->n_params, B, rows, state, v->seed, keep_thr, keep_scale, ga, slices);
            else
                hipLaunchKernelGGL(vae_bwd_dw_kernel<false>, dim3(v->dw_tiles, slices + grows), blk, smem, st, v->d_dw + (size_t)par * v->n_dw, v->n_dw,
                                   v->part, v->n_params, B, rows, state, v->seed, keep_thr, keep_scale, ga, slices);
        }
        if (g_vae_sync_each) (void)hipDeviceSynchronize();
    }
    // ---- optimiser ----
    vae_adam_args ad{};
    ad.params = v->params; ad.m = v->m; ad.v = v->v; ad.wt = v->wt; ad.wp = v->wp; ad.tpos = v->d_tpos; ad.tpos2 = v->d_tpos2;
    ad.part = v->part;
    ad.n_params = v->n_params; ad.slices = slices;
    ad.running = v->running; ad.stats = stats; ad.n_stats = v->n_stats; ad.rep_stride = (unsigned)v->n_stats; ad.bns = v->d_bns; ad.n_bn = (int)v->bns.size();
    ad.state = state; ad.state_next = state_next; ad.lr = v->lr; ad.beta1 = 0.9f; ad.beta2 = 0.999f; ad.eps = 1e-8f; ad.B = B;
    ad.K0 = v->d0; ad.data = gather_in_dw ? nullptr : d_data; ad.perm = d_perm; ad.batch = batch_next; ad.sums_part = v->sums_part; ad.sums = v->sums;
    ad.n_wg = (int)grid.x; ad.n_wg_loss = (int)(col_grid(v->outl.N, v->outl.K).x * col_grid(v->outl.N, v->outl.K).y); ad.w_cov = v->w_cov; ad.w_comp = v->w_comp; ad.w_kld = v->w_kld;
    if (multi)
        hipLaunchKernelGGL(vae_adam_kernel<true>, dim3((unsigned)((v->n_params + 255) / 256)), blk, 0, st, ad);
    else
        hipLaunchKernelGGL(vae_adam_kernel<false>, dim3((unsigned)((v->n_params + 255) / 256)), blk, 0, st, ad);
    HIP_TRY(hipGetLastError());
    return LRB_OK;
}

/* n_steps optimisation steps over consecutive batch_size slices of d_perm (row ids into
 * d_data [n_rows][cov+prof]); the loss sums of the steps accumulate in buffer 4. */
extern "C" int lrb_vae_train_dev(lrb_vae *v, const float *d_data, const int64_t *d_perm, uint32_t batch_size,
                                 uint32_t n_steps, int use_graph)
{
    ARG_TRY(v != nullptr);
    if (n_steps == 0) return LRB_OK;
    ARG_TRY(d_data && d_perm);
    ARG_TRY(batch_size >= 2 && (int)batch_size <= v->max_batch);
    hipStream_t st = v->ctx->stream;
    // position in the permutation restarts with every call; the first batch is fetched here,
    // every later one by the optimiser kernel of the step before it
    int par = (int)(v->host_steps & 1);
    {
        const vae_state init{v->host_steps, 0ull, (unsigned long long)batch_size * n_steps};
        HIP_TRY(hipMemcpyAsync(v->state + par, &init, sizeof init, hipMemcpyHostToDevice, st));
        unsigned blocks = (unsigned)(((size_t)batch_size * v->d0 + 255) / 256);
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(vae_gather_kernel, dim3(blocks), dim3(256), 0, st, d_data, (const long long *)d_perm, v->state + par,
                           v->batch + (size_t)par * v->max_batch * v->d0, (int)batch_size, v->d0);
        HIP_TRY(hipGetLastError());
    }
    if (!use_graph) {
        if (getenv("LRB_VAE_SYNC")) HIP_TRY(hipDeviceSynchronize());
        g_vae_sync_each = getenv("LRB_VAE_SYNC") && atoi(getenv("LRB_VAE_SYNC")) >= 2;
        for (uint32_t s = 0; s < n_steps; ++s, par ^= 1) {
            int rc = vae_enqueue_step(v, d_data, (const long long *)d_perm, (int)batch_size, st, par);
            if (rc != LRB_OK) return rc;
            ++v->host_steps;
        }
        return LRB_OK;
    }
    // one recorded step per (batch size, step parity, data, permutation) -- the pointers are baked in
    if (v->graph_data != d_data || v->graph_perm != d_perm) {
        for (hipGraphExec_t g : v->graph_exec) (void)hipGraphExecDestroy(g);
        v->graph_exec.clear();
        v->graph_B.clear();
        v->graph_data = d_data;
        v->graph_perm = d_perm;
    }
    // recorded: one step of either parity, and a block of VAE_GRAPH_BLOCK steps starting at parity 0
    // (an even count, so it can be replayed back to back): a graph launch costs ~9 us of its own
    auto get = [&](int key, int first_par, int steps, hipGraphExec_t *out) -> int {
        for (size_t i = 0; i < v->graph_B.size(); ++i)
            if (v->graph_B[i] == key) {
                *out = v->graph_exec[i];
                return LRB_OK;
            }
        hipGraph_t graph;
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipStreamBeginCapture(v->cap_stream, hipStreamCaptureModeThreadLocal));
        int rc = LRB_OK;
        for (int q = 0; q < steps && rc == LRB_OK; ++q)
            rc = vae_enqueue_step(v, d_data, (const long long *)d_perm, (int)batch_size, v->cap_stream, (first_par + q) & 1);
        hipError_t e = hipStreamEndCapture(v->cap_stream, &graph);
        if (rc != LRB_OK) return rc;
        HIP_TRY(e);
        HIP_TRY(hipGraphInstantiate(out, graph, nullptr, nullptr, 0));
        (void)hipGraphDestroy(graph);
        v->graph_B.push_back(key);
        v->graph_exec.push_back(*out);
        return LRB_OK;
    };
    constexpr int VAE_GRAPH_BLOCK = 8;
    uint32_t left = n_steps;
    while (left) {
        hipGraphExec_t exec = nullptr;
        int rc, took;
        if (par == 0 && left >= VAE_GRAPH_BLOCK) {
            rc = get((int)batch_size * 4 + 2, 0, VAE_GRAPH_BLOCK, &exec);
            took = VAE_GRAPH_BLOCK;
        } else {
            rc = get((int)batch_size * 4 + par, par, 1, &exec);
            took = 1;
        }
        if (rc != LRB_OK) return rc;
        HIP_TRY(hipGraphLaunch(exec, st));
        v->host_steps += (unsigned long long)took;
        left -= (uint32_t)took;
        par = (par + took) & 1;
    }
    return LRB_OK;
}

// running statistics -> the {sum, sum of squares} form the forward kernels read, for a count of 1
__global__ __launch_bounds__(256) void vae_eval_stats_kernel(const float *__restrict__ running, const vae_bn_desc *bns, int n_bn,
                                                             float *__restrict__ est)
{
    for (int q = 0; q < n_bn; ++q) {
        const vae_bn_desc d = bns[q];
        for (int i = blockIdx.x * 256 + threadIdx.x; i < d.n; i += gridDim.x * 256) {
            const float mean = running[d.run_off + i];
            est[d.stats_off + i] = mean;
            est[d.stats_off + d.n + i] = running[d.run_off + d.n + i] + mean * mean;
        }
    }
}

/* Latent means of n_rows rows (VAE.encode, ae_utils.py:141-161: eval mode -- running statistics,
 * no dropout, output = mu): d_mu[n_rows][latent] float32, input order.  Enqueued on the
 * context's stream. */
extern "C" int lrb_vae_encode_dev(lrb_vae *v, const float *d_data, uint64_t n_rows, float *d_mu)
{
    ARG_TRY(v != nullptr);
    if (n_rows == 0) return LRB_OK;
    ARG_TRY(d_data && d_mu);
    hipStream_t st = v->ctx->stream;
    const int nh = v->n_hidden;
    float *est = v->eval_stats;
    hipLaunchKernelGGL(vae_eval_stats_kernel, dim3(4), dim3(256), 0, st, v->running, v->d_bns, (int)v->bns.size(), est);
    auto bn_of = [&](int q) {
        const vae_bn_desc &d = v->bns[q];
        return vae_bn{est + d.stats_off, v->params + d.g_off, v->params + d.beta_off, (unsigned)v->n_stats};
    };
    const vae_bn none{nullptr, nullptr, nullptr, 0u};
    for (uint64_t r0 = 0; r0 < n_rows; r0 += (uint64_t)v->max_batch) {
        const int B = (int)(n_rows - r0 < (uint64_t)v->max_batch ? n_rows - r0 : (uint64_t)v->max_batch);
        const dim3 blk(256), grid((B + VT_M - 1) / VT_M);
        for (int i = 0; i < nh; ++i) {
            vae_fwd_args a{};
            a.in = i == 0 ? d_data + r0 * v->d0 : v->act_enc[i - 1];
            a.bn_in = i == 0 ? none : bn_of(i - 1);
            a.Wt = v->wt + v->enc[i].wt_off;
            a.bias = v->params + v->enc[i].b_off;
            a.out = v->act_enc[i];
            a.state = v->state;
            a.B = B; a.K = v->enc[i].K; a.N = v->enc[i].N; a.layer = i;
            a.keep_threshold = 0; a.keep_scale = 1.0f; a.eval = 1;
            hipLaunchKernelGGL((vae_fwd_kernel<VAE_ACT_BLOCK, false, 2>), grid, blk, vae_fwd_smem(a.K, 0), st, a);
        }
        vae_fwd_args a{};
        a.in = v->act_enc[nh - 1];
        a.bn_in = bn_of(nh - 1);
        a.Wt = v->wt + v->heads.wt_off;
        a.bias = v->params + v->heads.b_off;
        a.out = v->heads_out;
        a.state = v->state;
        a.B = B; a.K = v->heads.K; a.N = v->heads.N; a.layer = 100; a.eval = 1;
        hipLaunchKernelGGL((vae_fwd_kernel<VAE_ACT_HEADS, false, 2>), grid, blk, vae_fwd_smem(a.K, 0), st, a);
        HIP_TRY(hipGetLastError());
        // the mu half of [mu | logsigma]
        HIP_TRY(hipMemcpy2DAsync(d_mu + r0 * v->latent, (size_t)v->latent * 4, v->heads_out, (size_t)2 * v->latent * 4,
                                 (size_t)v->latent * 4, (size_t)B, hipMemcpyDeviceToDevice, st));
    }
    return LRB_OK;
}

/* test hook: copies of internal activations.  which: 0 eps [B][L], 1 z, 2 heads (mu | logsigma),
 * 3 dL/drecon, 10+i encoder block i output, 20+i decoder block i output,
 * 30 summed parameter gradient of the LAST step computed with `slices` partials. */
extern "C" int lrb_vae_debug_read(lrb_vae *v, int which, float *host, uint64_t count)
{
    ARG_TRY(v != nullptr && host != nullptr);
    HIP_TRY(hipStreamSynchronize(v->ctx->stream));
    const float *src = nullptr;
    if (which == 0) src = v->eps;
    else if (which == 1) src = v->z;
    else if (which == 2) src = v->heads_out;
    else if (which == 3) src = v->grad_out;
    else if (which >= 10 && which < 10 + v->n_hidden) src = v->act_enc[which - 10];
    else if (which >= 20 && which < 20 + v->n_hidden) src = v->act_dec[which - 20];
    else if (which >= 40 && which < 40 + v->n_hidden) src = v->dZ_dec[which - 40];
    else if (which >= 50 && which < 50 + v->n_hidden) src = v->dY_dec[which - 50];
    else if (which >= 60 && which < 60 + v->n_hidden) src = v->dZ_enc[which - 60];
    else if (which >= 70 && which < 70 + v->n_hidden) src = v->dY_enc[which - 70];
    else if (which == 80) src = v->stats;
    else if (which == 30) {
        ARG_TRY(count % v->n_params == 0 && count / v->n_params <= (uint64_t)v->max_slices);
        HIP_TRY(hipMemcpy(host, v->part, count * 4, hipMemcpyDeviceToHost));
        return LRB_OK;
    }
    ARG_TRY(src != nullptr);
    HIP_TRY(hipMemcpy(host, src, count * 4, hipMemcpyDeviceToHost));
    return LRB_OK;
}
