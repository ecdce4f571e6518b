// ubench_gather.hip -- what ONE random gather costs the memory system, by allocation kind (gfx950).
// K3 (coverage histograms) is one random byte / dword gather per window from a 512 MB map / 4 GiB table, and on
// ordinary device memory every gather leaves the L2 as a 128-byte line fill.  This asks whether
// memory the L2 does not cache (hipDeviceMallocUncached / hipDeviceMallocFinegrained) is fetched in smaller
// requests, and whether more of those fit through the XCD -> memory path per second.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_gather.hip -o gpurun_out/ubench_gather && gpurun_out/ubench_gather
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// every lane: `iters` rounds of 16 independent gathers of T at uniformly random element indices below `mask`+1
template <typename T, int POLICY>
__global__ __launch_bounds__(256) void gather(const T *__restrict__ tab, uint32_t mask, int iters, uint32_t *out)
{
    uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    uint32_t sink = 0;
    for (int it = 0; it < iters; ++it) {
        uint32_t idx[16];
        T v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) { x += 0x9e3779b9u; idx[j] = mix(x) & mask; }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (POLICY == 0) v[j] = tab[idx[j]];
            else v[j] = __builtin_nontemporal_load(tab + idx[j]);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) sink += (uint32_t)v[j];
    }
    if (sink == 0xdeadbeefu) out[0] = sink;
}

template <typename T, int POLICY>
static int run(const char *name, const void *tab, size_t bytes, uint32_t *out)
{
    const uint32_t mask = (uint32_t)(bytes / sizeof(T) - 1);
    const int grid = 256 * 8, iters = 256;
    const double gathers = (double)grid * 256 * iters * 16;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    gather<T, POLICY><<<grid, 256>>>((const T *)tab, mask, 8, out);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CHECK(hipEventRecord(e0));
        gather<T, POLICY><<<grid, 256>>>((const T *)tab, mask, iters, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%-34s %4zu B elems, %5zu MB: %8.2f ms  %6.1f G gathers/s\n", name, sizeof(T), bytes >> 20, best, gathers / best / 1e6);
    fflush(stdout);
    return 0;
}

int main()
{
    uint32_t *out;
    CHECK(hipMalloc(&out, 64));
    const size_t sizes[] = {512ull << 20, 64ull << 20, 2ull << 20};
    for (size_t bytes : sizes) {
        void *plain = nullptr, *unc = nullptr, *fine = nullptr;
        CHECK(hipMalloc(&plain, bytes));
        CHECK(hipMemset(plain, 1, bytes));
        hipError_t eu = hipExtMallocWithFlags(&unc, bytes, hipDeviceMallocUncached);
        hipError_t ef = hipExtMallocWithFlags(&fine, bytes, hipDeviceMallocFinegrained);
        if (eu != hipSuccess) { printf("uncached alloc: %s\n", hipGetErrorString(eu)); unc = nullptr; (void)hipGetLastError(); }
        if (ef != hipSuccess) { printf("finegrained alloc: %s\n", hipGetErrorString(ef)); fine = nullptr; (void)hipGetLastError(); }
        if (unc) CHECK(hipMemset(unc, 1, bytes));
        if (fine) CHECK(hipMemset(fine, 1, bytes));
        CHECK(hipDeviceSynchronize());
        if (run<uint8_t, 0>("hipMalloc, byte", plain, bytes, out)) return 1;
        if (run<uint32_t, 0>("hipMalloc, dword", plain, bytes, out)) return 1;
        if (run<uint8_t, 1>("hipMalloc, byte, nontemporal", plain, bytes, out)) return 1;
        if (unc) {
            if (run<uint8_t, 0>("uncached, byte", unc, bytes, out)) return 1;
            if (run<uint32_t, 0>("uncached, dword", unc, bytes, out)) return 1;
            if (run<uint8_t, 1>("uncached, byte, nontemporal", unc, bytes, out)) return 1;
        }
        if (fine) {
            if (run<uint8_t, 0>("finegrained, byte", fine, bytes, out)) return 1;
            if (run<uint32_t, 0>("finegrained, dword", fine, bytes, out)) return 1;
        }
        CHECK(hipFree(plain));
        if (unc) CHECK(hipFree(unc));
        if (fine) CHECK(hipFree(fine));
    }
    return 0;
}
