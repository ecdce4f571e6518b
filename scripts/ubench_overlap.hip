// Do a CU's VALU work and its LDS atomics overlap when every wave does both?  (Round 5: the part kernel's counters say
// VALU 49 % + LDS 49 % of its cycles, and its time is their sum.)  One workgroup of 16 waves a CU (128 KB of LDS, as the part
// kernel), every lane a stream of pseudo-random window codes; per window
//   V: the part kernel's arithmetic (~24 integer instructions, results kept alive)
//   L: one returning 64-bit LDS atomic on one of 256 cells + one 4-byte store into a 128 KB ring area
// Kernels: V only, L only (addresses from a cheap counter), V + L as the part kernel interleaves them (four windows' atomics in
// flight while the next four are made), and V + L with NO dependence of L on V's result.  Time per launch, and V + L beside the sum.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_overlap.hip -o scripts/bin/ubench_overlap && scripts/bin/ubench_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef __attribute__((address_space(3))) uint32_t l32;
typedef __attribute__((address_space(3))) unsigned long long l64;

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12; x *= 0x297a2d39u; x ^= x >> 15; return x; }

// ~24 VALU instructions on a code word pair (alignbit, masks, selects, bit-field moves): the part kernel's per-window arithmetic
__device__ __forceinline__ uint32_t work(uint32_t a, uint32_t b, uint32_t i)
{
    uint32_t v = __builtin_amdgcn_alignbit(a, b, (2 * i) & 31) & 0x3FFFFFFFu;
    uint32_t r = __builtin_amdgcn_alignbit(~b, ~a, (30 - 2 * i) & 31) & 0x3FFFFFFFu;
    uint32_t x = (v & 0x8000u) ? r : v;
    uint32_t e = (((x >> 1) & ~0x7FFFu) | (x & 0x7FFFu)) & 0x1FFFFFu;
    x ^= e * 0x9E3779B1u;            // a few more dependent operations
    x = (x >> 7) | (x << 25);
    x += e;
    x ^= x >> 11;
    x *= 0x85EBCA6Bu;
    x ^= x >> 13;
    x += a;
    x ^= b >> 3;
    return x;
}

template <int MODE> // 0: V only, 1: L only, 2: V + L (L's address from V), 3: V + L independent, 4: as 2 + two barriers per 16 windows,
                    // 5: as 4 + the flush's LDS reads and global stores (32 bytes a lane per 8 windows) between the barriers
__global__ __launch_bounds__(1024) void k(uint32_t *out, uint32_t iters, uint32_t *sink)
{
    extern __shared__ uint32_t smem[];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < 2048 / 4 * 2; i += 1024) smem[i] = 0;
    __syncthreads();
    uint32_t a = mix(tid * 7919u + blockIdx.x), b = mix(a), acc = 0, ctr = a;
    for (uint32_t it = 0; it < iters; ++it) {
        uint32_t x[4];
        unsigned long long r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (MODE != 1) x[j] = work(a, b, it * 4 + j);
            else x[j] = ctr += 0x9E3779B1u;
        }
        if (MODE != 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t src = MODE == 3 ? (ctr += 0x9E3779B1u) : x[j];
                r[j] = __hip_atomic_fetch_add((l64 *)((src >> 19) & 0x7F8u), 4ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t src = MODE == 3 ? ctr + j : x[j];
                *(l32 *)(2048u + (((src >> 13) & 0x1FE00u) | ((uint32_t)r[j] & 0x1FCu))) = x[j];
                acc += (uint32_t)(r[j] >> 32);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc ^= x[j];
        if (MODE >= 4 && (it & 3u) == 3u) {
            __syncthreads();
            if (MODE == 5) { // sixteen windows a lane were appended: 64 bytes a lane go out (two lines of a slice per four lanes)
                const uint32_t s_ = tid >> 2, q = (tid & 3u) * 32u;
                const unsigned long long tf = *(l64 *)(8u * (s_ & 255u));
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    const uint32_t la = 2048u + (s_ & 255u) * 512u + (((uint32_t)tf + 128u * k2 + q) & 0x1FFu);
                    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
                    const v4 v0 = *(__attribute__((address_space(3))) v4 *)la, v1 = *(__attribute__((address_space(3))) v4 *)(la + 16);
                    v4 *d = (v4 *)(sink + ((size_t)blockIdx.x * 512 + (it >> 2) % 512) * 16384 + (tid * 2 + k2) * 8);
                    d[0] = v0;
                    d[1] = v1;
                }
            }
            __syncthreads();
        }
        a += 0x632BE5ABu;
        b ^= a >> 5;
    }
    out[blockIdx.x * 1024 + tid] = acc;
}

int main()
{
    CK(hipSetDevice(0));
    uint32_t *out;
    CK(hipMalloc((void **)&out, 256 * 1024 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const uint32_t iters = 4096; // x 4 windows x 1024 threads x 256 CUs = 4.3e9 windows
    const size_t lds = 2048 + 131072;
    CK(hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute((const void *)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute((const void *)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute((const void *)k<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute((const void *)k<5>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    uint32_t *sink; // 256 workgroups x 512 flushes x 64 KB = 8 GB, written round and round (whole 128-byte lines, four lanes a line)
    CK(hipMalloc((void **)&sink, (size_t)256 * 512 * 16384 * 4));
    float ms[6] = {0, 0, 0, 0, 0, 0};
    for (int rep = 0; rep < 3; ++rep)
        for (int m = 0; m < 6; ++m) {
            CK(hipEventRecord(e0));
            if (m == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(1024), lds, 0, out, iters, sink);
            if (m == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(1024), lds, 0, out, iters, sink);
            if (m == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(1024), lds, 0, out, iters, sink);
            if (m == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(1024), lds, 0, out, iters, sink);
            if (m == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(1024), lds, 0, out, iters, sink);
            if (m == 5) hipLaunchKernelGGL(k<5>, dim3(256), dim3(1024), lds, 0, out, iters, sink);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms[m], e0, e1));
        }
    printf("4.3e9 windows, one workgroup of 16 waves a CU:\n  arithmetic only            %.3f ms\n  LDS atomic + store only    %.3f ms\n"
           "  both, LDS address from the arithmetic   %.3f ms   (sum %.3f, max %.3f)\n  both, independent          %.3f ms\n"
           "  both + two barriers per 16 windows a lane   %.3f ms\n  ... + the flush's LDS reads and stores      %.3f ms\n",
           ms[0], ms[1], ms[2], ms[0] + ms[1], ms[0] > ms[1] ? ms[0] : ms[1], ms[3], ms[4], ms[5]);
    return 0;
}
