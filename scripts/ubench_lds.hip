// ubench_lds.hip -- LDS atomic throughput microbenchmark for the K1 design (gfx950).
// Measures CU-cycles per ds_add_u32 wave-instruction for the access patterns K1 can use.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_lds.hip -o gpurun_out/ubench_lds && gpurun_out/ubench_lds
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// MODE 0: [bin][32 subs] u32, inc 1            (conflict free)
// MODE 1: [bin][16 subs]                        (2 lanes of a 32-group may share a bank)
// MODE 2: [bin][8 subs]
// MODE 3: packed u8: word = bin>>2, inc = 1<<(8*(bin&3)), 32 subs
// MODE 4: all lanes same sub (bin*1): heavy conflicts
// MODE 5: VALU only (no ds_add) -- cost of the address arithmetic
// MODE 6: ds_add with wave-uniform address per lane == lane (no VALU): pure LDS issue
template <int MODE, int BINS>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint32_t seed)
{
    extern __shared__ uint32_t smem[];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int SUBS = MODE == 1 ? 16 : MODE == 2 ? 8 : MODE == 4 ? 1 : 32;
    constexpr int WORDS = MODE == 3 ? (BINS / 4) * 32 : BINS * SUBS;
    uint32_t *h = smem + wave * WORDS;
    for (int i = lane; i < WORDS; i += 64) h[i] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    uint32_t x = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    const uint32_t sub = lane & (SUBS - 1);
    uint32_t sink = 0;
    for (int it = 0; it < iters; ++it) {
        x = x * 1664525u + 1013904223u;
        uint32_t w = x;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const uint32_t bin = (w >> (p)) & (BINS - 1);
            if (MODE == 5) {
                sink += bin * SUBS + sub;
            } else if (MODE == 6) {
                atomicAdd(&h[lane], 1u);
            } else if (MODE == 3) {
                atomicAdd(&h[(bin >> 2) * 32 + sub], 1u << (8 * (bin & 3)));
            } else {
                atomicAdd(&h[bin * SUBS + sub], 1u);
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    uint32_t s = sink;
    for (int i = lane; i < WORDS; i += 64) s += h[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int BINS>
int run(const char *name, int blocks_per_cu, int pad_lds_kb)
{
    constexpr int SUBS = MODE == 1 ? 16 : MODE == 2 ? 8 : MODE == 4 ? 1 : 32;
    constexpr int WORDS = MODE == 3 ? (BINS / 4) * 32 : BINS * SUBS;
    size_t smem = (size_t)4 * WORDS * 4;
    if (pad_lds_kb * 1024 > (int)smem) smem = pad_lds_kb * 1024;
    CHECK(hipFuncSetAttribute((const void *)k<MODE, BINS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const int grid = 256 * blocks_per_cu, iters = 2000;
    uint32_t *out;
    CHECK(hipMalloc(&out, (size_t)grid * 256 * 4));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<MODE, BINS>), dim3(grid), dim3(256), smem, 0, out, 10, 1u);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL((k<MODE, BINS>), dim3(grid), dim3(256), smem, 0, out, iters, (uint32_t)r + 2);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    // wave-instructions per CU: blocks_per_cu * 4 waves * iters * 16
    const double winstr = (double)blocks_per_cu * 4 * iters * 16;
    const double ns_per = best * 1e6 / winstr;
    printf("%-34s bins=%4d waves/CU=%2d lds/blk=%6zu  %.3f ms  %.2f ns per wave-instr per CU (= %.2f cyc @2.4GHz, %.1f lane-ops/clk)\n",
           name, BINS, blocks_per_cu * 4, smem, best, ns_per, ns_per * 2.4, 64.0 / (ns_per * 2.4));
    CHECK(hipFree(out));
    return 0;
}

int main()
{
    for (int bpc : {1, 2, 4, 8}) {
        run<0, 64>("u32 [bin][32]", bpc, 0);
    }
    run<0, 256>("u32 [bin][32] 256 bins", 1, 0);
    for (int bpc : {2, 4, 8}) run<1, 64>("u32 [bin][16]", bpc, 0);
    for (int bpc : {2, 4, 8}) run<2, 64>("u32 [bin][8]", bpc, 0);
    for (int bpc : {2, 4, 8}) run<3, 256>("u8 packed [bin/4][32] var inc", bpc, 0);
    for (int bpc : {2, 4}) run<4, 64>("u32 [bin] shared (conflicts)", bpc, 0);
    for (int bpc : {2, 4, 8}) run<5, 64>("VALU only", bpc, 0);
    for (int bpc : {1, 2, 4, 8}) run<6, 64>("ds_add fixed addr (pure issue)", bpc, 0);
    return 0;
}
