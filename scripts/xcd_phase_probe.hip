// xcd_phase_probe.hip -- what ONE PHASE of a VAE step costs in a kernel written for one XCD, before
// anybody writes that kernel.  The model said: barrier 0.3 + tile from the L2 1.0 + into LDS 0.2 + 128 MFMAs per wave 1.7
// + epilogue 1.2 = 4.4 us.  Here the 32 workgroups of one XCD (one per CU, elected by HW_REG_XCC_ID as in
// xcd_sync_probe.hip) run a chain of PHASES layers of a 1024-row batch through 128 x 128 fp32 weights:
//   workgroup w owns rows 32 w .. 32 w + 31 through the whole chain (two 16-row MFMA tiles over ONE staging of the weights);
//   the next phase's weights (64 KB, shared by everybody: L2 hits) are requested BEFORE the barrier and written to the other
//   LDS buffer after the MFMA loop (they wait in registers meanwhile);
//   the column sums of a phase (BatchNorm statistics) travel as 32 slabs of 2 x 128 floats through the L2: plain stores,
//   L1-bypassing loads, fixed summation order -- no atomics;
//   activations: plain stores, L1-bypassing 16-byte loads, BatchNorm affine applied on the way into LDS;
//   epilogue: bias, LeakyReLU, a dropout hash per element (as the real step's), store, column sums.
// Timed per phase with s_memrealtime; the same chain run as one LAUNCH per phase (256 CUs free, stream order) gives the
// bits to compare with and the launch-per-phase time of this very body.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/xcd_phase_probe.hip -o /tmp/xcd_phase_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int B = 1024, W = 128, ROWS = 32, NWG = B / ROWS;   // 32 workgroups
constexpr int LDA = W + 1, LDB = W + 16;                      // LDS strides (A: 129, B: 144 = the real kernels' VT_NS)
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t xcc_id()
{
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}

struct ctl_t {
    uint32_t arrivals[16];
    uint32_t pad[16];
    uint32_t flags[64];
    uint32_t participants, elected, timeouts, pad2;
};

__device__ __forceinline__ uint32_t hash32(uint32_t a, uint32_t b)
{
    uint32_t x = a * 0x9E3779B1u ^ b;
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}

// 16-byte loads that bypass the CU's L1 (data another CU of this XCD wrote during the launch): buffer load, aux = sc1
__device__ __forceinline__ float4 ld4_sc1(__amdgpu_buffer_rsrc_t rs, uint32_t byte_off)
{
    const v4u_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, 16);
    float4 f;
    f.x = __uint_as_float(v.x); f.y = __uint_as_float(v.y); f.z = __uint_as_float(v.z); f.w = __uint_as_float(v.w);
    return f;
}

struct args_t {
    float *act[2];        // [B][W] ping-pong
    const float *wts;     // [phases][W k][W n] (k-major: row k holds the 128 outputs' weights)
    const float *bias;    // [phases][W]
    float *slabs;         // [2][NWG][2 W]: column sums / sums of squares of a phase's output, per workgroup, ping-pong
    ctl_t *ctl;
    uint64_t *stamps;     // [phases + 1] of workgroup 0
    int phases, persistent, phase0;   // (launch-per-phase mode: phases == 1, phase0 = which)
};

// one phase for this workgroup's 32 rows.  Wl: this phase's weights, already in LDS.
template <bool SC1>
__device__ __forceinline__ void phase_body(const args_t &a, int p, int rank, float *As, const float *Wl, float *tab, float *red)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *in = a.act[p & 1];
    float *out = a.act[(p + 1) & 1];
    // ---- BatchNorm table of the previous phase's output: thread t sums column t & 127 over 16 of the 32 slabs ----
    if (p > 0) {
        const float *sl = a.slabs + (size_t)((p - 1) & 1) * NWG * 2 * W;
        const int c = tid & 127, half = tid >> 7;
        float s = 0.0f, q = 0.0f;
        float sv[16], qv[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float *src = sl + (size_t)(half * 16 + j) * 2 * W;
            if (SC1) {
                sv[j] = __hip_atomic_load(src + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                qv[j] = __hip_atomic_load(src + W + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                sv[j] = src[c];
                qv[j] = src[W + c];
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) { s += sv[j]; q += qv[j]; }
        red[half * 2 * W + c] = s;
        red[half * 2 * W + W + c] = q;
    }
    // ---- the activation tile, 16-byte loads: thread t takes row t / 8, columns 16 (t % 8) .. +15 ----
    const int r = tid >> 3, c0 = (tid & 7) * 16;
    float4 x[4];
    const float *src = in + (size_t)(rank * ROWS + r) * W + c0;
    const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, B * W * 4, 0x00020000);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        x[j] = SC1 ? ld4_sc1(irs, (uint32_t)(((rank * ROWS + r) * W + c0 + 4 * j) * 4)) : *reinterpret_cast<const float4 *>(src + 4 * j);
    __syncthreads();
    if (tid < W) {
        float sc = 1.0f, sh = 0.0f;
        if (p > 0) {
            const float s = red[tid] + red[2 * W + tid], q = red[W + tid] + red[3 * W + tid];
            const float mean = s * (1.0f / B);
            float var = q * (1.0f / B) - mean * mean;
            var = var > 0.0f ? var : 0.0f;
            sc = rsqrtf(var + 1e-5f);
            sh = -mean * sc;
        }
        tab[tid] = sc;
        tab[W + tid] = sh;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float v[4] = {x[j].x, x[j].y, x[j].z, x[j].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = c0 + 4 * j + e;
            As[r * LDA + c] = fmaf(v[e], tab[c], tab[W + c]);
        }
    }
    __syncthreads();
    // ---- 32 rows x 128 columns: wave w owns columns 32 w .. +31: two row tiles x two column tiles ----
    v4f_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = v4f_t{0.0f, 0.0f, 0.0f, 0.0f};
    const float *ap = As + (lane & 15) * LDA + (lane >> 4);
    const float *bp = Wl + (lane >> 4) * LDB + wave * 32 + (lane & 15);
#pragma unroll 4
    for (int k = 0; k < W; k += 4) {
        const float a0 = ap[k], a1 = ap[16 * LDA + k];
        const float b0 = bp[k * LDB], b1 = bp[k * LDB + 16];
        acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    // ---- epilogue: lane holds acc[i][j][e] = C[row 16 i + 4 (lane / 16) + e][col 32 wave + 16 j + lane % 16] ----
    const float *bias = a.bias + (size_t)p * W;
    float cs[2] = {0.0f, 0.0f}, cq[2] = {0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = 32 * wave + 16 * j + (lane & 15);
        const float bv = bias[col];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = rank * ROWS + 16 * i + 4 * (lane >> 4) + e;
                float v = acc[i][j][e] + bv;
                v = v > 0.0f ? v : 0.01f * v;
                const uint32_t h = hash32((uint32_t)p * 7919u + 17u, (uint32_t)(row * W + col));
                v = h >= 429496730u ? v * (1.0f / 0.9f) : 0.0f;   // dropout 0.1
                out[(size_t)row * W + col] = v;
                cs[j] += v;
                cq[j] += v * v;
            }
    }
    // column sums over this workgroup's 32 rows: the four lane groups (lane / 16) hold different rows of a column
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        cs[j] += __shfl_xor(cs[j], 16, 64); cs[j] += __shfl_xor(cs[j], 32, 64);
        cq[j] += __shfl_xor(cq[j], 16, 64); cq[j] += __shfl_xor(cq[j], 32, 64);
    }
    if (lane < 16) {
        float *sl = a.slabs + ((size_t)(p & 1) * NWG + rank) * 2 * W;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = 32 * wave + 16 * j + lane;
            sl[col] = cs[j];
            sl[W + col] = cq[j];
        }
    }
}

// weights of phase p: global -> registers (issue), registers -> LDS (commit): 128 x 128 floats, 16 float4 per thread
struct wregs_t { float4 v[16]; };
__device__ __forceinline__ void w_issue(wregs_t &w, const float *wts, int p)
{
    const float4 *src = reinterpret_cast<const float4 *>(wts + (size_t)p * W * W) + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 16; ++u) w.v[u] = src[u * 256];
}
__device__ __forceinline__ void w_commit(const wregs_t &w, float *Wl)
{
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int q = u * 256 + threadIdx.x, k = q >> 5, c = (q & 31) * 4;
        *reinterpret_cast<float4 *>(Wl + k * LDB + c) = w.v[u];
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void chain_kernel(args_t a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Wl[2] = {lds, lds};   // ONE buffer: the next phase's weights wait in registers until this phase's MFMAs are through
    float *As = lds + W * LDB;
    float *tab = As + ROWS * LDA + 3;
    tab = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(tab) + 15) & ~(uintptr_t)15);
    float *red = tab + 2 * W;
    __shared__ uint32_t s_rank, s_go;
    const int tid = threadIdx.x;
    if (!a.persistent) {   // one launch per phase: workgroup index = rank, weights staged in line
        wregs_t w;
        w_issue(w, a.wts, a.phase0);
        w_commit(w, Wl[0]);
        __syncthreads();
        phase_body<false>(a, a.phase0, (int)blockIdx.x, As, Wl[0], tab, red);
        return;
    }
    const uint32_t me = xcc_id();
    if (tid == 0) {
        s_rank = atomicAdd(&a.ctl->arrivals[me], 1u);
        s_go = me == 0u ? 1u : 0u;
    }
    __syncthreads();
    if (!s_go || s_rank >= (uint32_t)NWG) return;
    const int rank = (int)s_rank;
    if (tid == 0) {   // everybody placed?
        for (int spin = 0; spin < (1 << 22); ++spin) {
            uint32_t tot = 0;
            for (int x = 0; x < 8; ++x) tot += __hip_atomic_load(&a.ctl->arrivals[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tot == gridDim.x) break;
            __builtin_amdgcn_s_sleep(2);
        }
        if (rank == 0) a.ctl->participants = __hip_atomic_load(&a.ctl->arrivals[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    auto barrier = [&](int r) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_store(&a.ctl->flags[rank], (uint32_t)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (tid < 64) {
            bool ok = false;
            for (int spin = 0; spin < (1 << 20); ++spin) {
                const uint32_t f = tid < NWG ? __hip_atomic_load(&a.ctl->flags[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (uint32_t)r;
                if (__all((int)(f >= (uint32_t)r))) { ok = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (!ok && tid == 0) atomicAdd(&a.ctl->timeouts, 1u);
        }
        __syncthreads();
    };
    wregs_t w;
    w_issue(w, a.wts, 0);
    w_commit(w, Wl[0]);
    __syncthreads();
    barrier(1);
    if (rank == 0 && tid == 0) a.stamps[0] = __builtin_amdgcn_s_memrealtime();
    for (int p = 0; p < a.phases; ++p) {
        if (p + 1 < a.phases) w_issue(w, a.wts, p + 1);          // next phase's weights: in flight during this phase
        phase_body<true>(a, p, rank, As, Wl[p & 1], tab, red);
        if (p + 1 < a.phases) {                                  // this phase's MFMAs are through (all waves): overwrite
            __syncthreads();
            w_commit(w, Wl[(p + 1) & 1]);
        }
        barrier(p + 2);
        if (rank == 0 && tid == 0) a.stamps[p + 1] = __builtin_amdgcn_s_memrealtime();
    }
}

int main(int argc, char **argv)
{
    const int phases = argc > 1 ? atoi(argv[1]) : 10;
    const int reps = argc > 2 ? atoi(argv[2]) : 50;
    const size_t smem = (size_t)(W * LDB + ROWS * LDA + 8 + 2 * W + 4 * W) * 4;
    float *act[2][2], *wts, *bias, *slabs[2];
    ctl_t *ctl;
    uint64_t *stamps;
    std::vector<float> h_in((size_t)B * W), h_w((size_t)phases * W * W), h_b((size_t)phases * W);
    srand(1);
    for (auto &v : h_in) v = (float)rand() / RAND_MAX;
    for (auto &v : h_w) v = ((float)rand() / RAND_MAX - 0.5f) * 0.2f;
    for (auto &v : h_b) v = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
    for (int m = 0; m < 2; ++m) {
        for (int i = 0; i < 2; ++i) CK(hipMalloc(&act[m][i], (size_t)B * W * 4));
        CK(hipMalloc(&slabs[m], (size_t)2 * NWG * 2 * W * 4));
    }
    CK(hipMalloc(&wts, h_w.size() * 4)); CK(hipMalloc(&bias, h_b.size() * 4));
    CK(hipMalloc(&ctl, sizeof(ctl_t))); CK(hipMalloc(&stamps, 8 * 64));
    CK(hipMemcpy(wts, h_w.data(), h_w.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, h_b.data(), h_b.size() * 4, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void *)chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int n_cu = 0;
    CK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0));
    printf("%d phases of a 1024 x 128 batch through 128 x 128 fp32 weights (BatchNorm sums as slabs), LDS %zu KB per workgroup\n", phases, smem >> 10);
    // ---- launch per phase (32 workgroups anywhere, stream order) ----
    float ms_launch = 0.0f;
    for (int rep = 0; rep <= reps; ++rep) {
        CK(hipMemcpy(act[0][0], h_in.data(), h_in.size() * 4, hipMemcpyHostToDevice));
        if (rep == 1) CK(hipEventRecord(e0, 0));
        for (int p = 0; p < phases; ++p) {
            args_t a{{act[0][0], act[0][1]}, wts, bias, slabs[0], ctl, stamps, 1, 0, p};
            hipLaunchKernelGGL(chain_kernel, dim3(NWG), dim3(256), smem, 0, a);
        }
    }
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms_launch, e0, e1));
    // (the copies in the timed loop cost the same in both forms; they are subtracted by timing them alone)
    float ms_copy = 0.0f;
    CK(hipEventRecord(e0, 0));
    for (int rep = 0; rep < reps; ++rep) CK(hipMemcpy(act[1][0], h_in.data(), h_in.size() * 4, hipMemcpyHostToDevice));   // (not the buffers compared below)
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms_copy, e0, e1));
    printf("one launch per phase, 32 workgroups: %.2f us per phase (stream order, plain launches)\n", (ms_launch - ms_copy) * 1e3 / (reps * phases));
    // ---- one persistent launch on XCC 0 ----
    std::vector<uint64_t> h_st(64);
    double sum_phase[64] = {0};
    int good = 0;
    uint32_t participants = 0, timeouts = 0;
    for (int rep = 0; rep <= reps; ++rep) {
        CK(hipMemcpy(act[1][0], h_in.data(), h_in.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemset(ctl, 0, sizeof(ctl_t)));
        args_t a{{act[1][0], act[1][1]}, wts, bias, slabs[1], ctl, stamps, phases, 1, 0};
        hipLaunchKernelGGL(chain_kernel, dim3(n_cu), dim3(256), smem > (84u << 10) ? smem : (84u << 10), 0, a);
        CK(hipDeviceSynchronize());
        ctl_t hc;
        CK(hipMemcpy(&hc, ctl, sizeof hc, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h_st.data(), stamps, 8 * 64, hipMemcpyDeviceToHost));
        participants = hc.participants; timeouts += hc.timeouts;
        if (rep == 0 || hc.participants != (uint32_t)NWG) continue;
        ++good;
        for (int p = 0; p < phases; ++p) sum_phase[p] += (double)(h_st[p + 1] - h_st[p]) / 100.0;
    }
    printf("persistent on one XCD: %u participants, %u barrier time-outs, %d timed launches\n  us per phase:", participants, timeouts, good);
    double tot = 0;
    for (int p = 0; p < phases; ++p) { printf(" %.2f", sum_phase[p] / (good ? good : 1)); tot += sum_phase[p] / (good ? good : 1); }
    printf("  | mean %.2f us\n", tot / phases);
    // ---- same bits? ----
    std::vector<float> r0((size_t)B * W), r1((size_t)B * W);
    CK(hipMemcpy(r0.data(), act[0][phases & 1], r0.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(r1.data(), act[1][phases & 1], r1.size() * 4, hipMemcpyDeviceToHost));
    size_t diff = 0;
    double checksum = 0;
    for (size_t i = 0; i < r0.size(); ++i) { diff += memcmp(&r0[i], &r1[i], 4) != 0; checksum += r0[i]; }
    printf("outputs of the two forms: %zu of %zu words differ (checksum %.6f)\n", diff, r0.size(), checksum);
    return 0;
}
