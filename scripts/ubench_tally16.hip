// What would K2's tally gain from reading a u16 OFFSET PLANE instead of the 4-byte list entries?  
// Synthetic lists with the product's structure at 400 k x 10 kb: 256 groups x 16,384 buckets, a run of ~954 entries per (group,
// bucket), entry = {read : 11 | offset in the slice : 21} (low 15 bits = offset in the bucket); a workgroup per bucket walks
// every group's run with the bucket's 2^15 counters in LDS and adds them to H as one coalesced read-modify-write -- the walk of
// wl_tally_kernel (pieces of sixteen loads a lane, the next piece's loads in flight while this one's are tallied), once over the
// u32 entries and once over a u16 plane of the offsets alone (two offsets a loaded dword; runs start on even indices here,
// which a product kernel would have to arrange or patch).
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_tally16.hip -o scripts/bin/ubench_tally16 && scripts/bin/ubench_tally16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x)                                                          \
    do {                                                               \
        hipError_t e_ = (x);                                           \
        if (e_ != hipSuccess) {                                        \
            printf("%s: %s\n", #x, hipGetErrorString(e_));             \
            return 1;                                                  \
        }                                                              \
    } while (0)

constexpr uint32_t GROUPS = 256, BUCKETS = 16384, RUN = 954; // 256 x 16384 x 954 = 4.0e9 entries

__device__ __forceinline__ uint32_t mix(uint64_t x)
{
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return (uint32_t)x;
}

// run lengths RUN +- 64, rounded to even; starts[g][b] (entries from the group's base)
__global__ void make_bounds(uint32_t *bounds)
{
    const uint32_t g = blockIdx.x;
    if (threadIdx.x) return;
    uint32_t at = 0;
    for (uint32_t b = 0; b < BUCKETS; ++b) {
        bounds[(uint64_t)g * (BUCKETS + 1) + b] = at;
        at += (RUN - 64 + (mix(((uint64_t)g << 32) | b) & 127u)) & ~1u;
    }
    bounds[(uint64_t)g * (BUCKETS + 1) + BUCKETS] = at;
}

__global__ void make_lists(uint32_t *lists, uint16_t *plane, uint64_t stride)
{
    const uint64_t n = (uint64_t)GROUPS * stride;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = mix(i * 0x9E3779B97F4A7C15ull + 12345);
        lists[i] = r;                          // {read : 11 | offset : 21}: all bits random
        plane[i] = (uint16_t)(r & 0x7FFFu);    // the offset in the bucket
    }
}

template <bool PLANE>
__global__ __launch_bounds__(1024) void tally(const uint32_t *__restrict__ lists, const uint16_t *__restrict__ plane,
                                               const uint32_t *__restrict__ bounds, uint64_t stride, uint32_t *__restrict__ half)
{
    extern __shared__ uint32_t hist[];
    const uint32_t b = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint32_t i = tid; i < 8192u; i += 1024) reinterpret_cast<uint4 *>(hist)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    // entries per piece: 1024 (u32) / 2048 (u16 plane: two a dword)
    constexpr uint32_t PER = PLANE ? 2048u : 1024u;
    auto ask = [&](const void *src, uint32_t len, uint32_t *e) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(src), 0, (int)(len * (PLANE ? 2u : 4u)), 0x00020000);
#pragma unroll
        for (int q = 0; q < 16; ++q) e[q] = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(lane * 4u), q * 256, 0);
    };
    auto add = [&](uint32_t len, const uint32_t *e) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (PLANE) {
                const uint32_t i0 = (q * 64 + lane) * 2;
                if (i0 < len) atomicAdd(&hist[e[q] & 0x7FFFu], 1u);
                if (i0 + 1 < len) atomicAdd(&hist[(e[q] >> 16) & 0x7FFFu], 1u);
            } else if (q * 64 + lane < len)
                atomicAdd(&hist[e[q] & 0x7FFFu], 1u);
        }
    };
    uint32_t ea[16], eb[16];
    for (uint32_t g = wave; g < GROUPS; g += 16) {
        const uint32_t s0 = bounds[(uint64_t)g * (BUCKETS + 1) + b], s1 = bounds[(uint64_t)g * (BUCKETS + 1) + b + 1];
        const uint64_t base = (uint64_t)g * stride + s0;
        uint32_t off = 0;
        const uint32_t total = s1 - s0;
        auto piece = [&](const void *&src, uint32_t &len) {
            src = PLANE ? (const void *)(plane + base + off) : (const void *)(lists + base + off);
            len = off < total ? (total - off < PER ? total - off : PER) : 0;
            off += PER;
        };
        const void *sa, *sb;
        uint32_t la, lb;
        piece(sa, la);
        ask(sa, la, ea);
        while (la) {
            piece(sb, lb);
            ask(sb, lb, eb);
            add(la, ea);
            if (!lb) break;
            piece(sa, la);
            ask(sa, la, ea);
            add(lb, eb);
        }
    }
    __syncthreads();
    uint4 *t = reinterpret_cast<uint4 *>(half + ((uint64_t)b << 15));
    for (uint32_t i = tid; i < 8192u; i += 1024) {
        const uint4 v = reinterpret_cast<const uint4 *>(hist)[i];
        uint4 o = t[i];
        o.x += v.x;
        o.y += v.y;
        o.z += v.z;
        o.w += v.w;
        t[i] = o;
    }
}

int main()
{
    CK(hipSetDevice(0));
    uint32_t *bounds, *lists, *half_a, *half_b;
    uint16_t *plane;
    CK(hipMalloc((void **)&bounds, sizeof(uint32_t) * GROUPS * (BUCKETS + 1)));
    hipLaunchKernelGGL(make_bounds, dim3(GROUPS), dim3(64), 0, 0, bounds);
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> last(1);
    uint64_t stride = 0;
    for (uint32_t g = 0; g < GROUPS; ++g) {
        CK(hipMemcpy(last.data(), bounds + (uint64_t)g * (BUCKETS + 1) + BUCKETS, 4, hipMemcpyDeviceToHost));
        if (last[0] > stride) stride = last[0];
    }
    stride = (stride + 63) & ~63ull;
    const uint64_t n = (uint64_t)GROUPS * stride;
    printf("%u groups x %u buckets, %.3f G entries (%.1f GB as u32, %.1f GB as a u16 plane)\n", GROUPS, BUCKETS, n / 1e9, n * 4 / 1e9, n * 2 / 1e9);
    CK(hipMalloc((void **)&lists, n * 4 + 64));
    CK(hipMalloc((void **)&plane, n * 2 + 64));
    CK(hipMalloc((void **)&half_a, 4ull << 29));
    CK(hipMalloc((void **)&half_b, 4ull << 29));
    CK(hipMemset(half_a, 0, 4ull << 29));
    CK(hipMemset(half_b, 0, 4ull << 29));
    hipLaunchKernelGGL(make_lists, dim3(4096), dim3(256), 0, 0, lists, plane, stride);
    CK(hipDeviceSynchronize());
    CK(hipFuncSetAttribute((const void *)tally<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void *)tally<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        float ms[2];
        for (int v = 0; v < 2; ++v) {
            CK(hipEventRecord(e0));
            if (v == 0) hipLaunchKernelGGL(tally<false>, dim3(BUCKETS), dim3(1024), 131072, 0, lists, plane, bounds, stride, half_a);
            else hipLaunchKernelGGL(tally<true>, dim3(BUCKETS), dim3(1024), 131072, 0, lists, plane, bounds, stride, half_b);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms[v], e0, e1));
        }
        printf("rep %d: tally over u32 entries %.3f ms (%.2f TB/s of list bytes), over the u16 offset plane %.3f ms (%.2f TB/s)\n", rep, ms[0],
               n * 4 / ms[0] / 1e9, ms[1], n * 2 / ms[1] / 1e9);
    }
    // the two halves must agree (the same offsets were tallied the same number of times)
    std::vector<uint32_t> a(1 << 20), b(1 << 20);
    CK(hipMemcpy(a.data(), half_a + (123ull << 20), 4 << 20, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), half_b + (123ull << 20), 4 << 20, hipMemcpyDeviceToHost));
    uint64_t diff = 0, sum = 0;
    for (size_t i = 0; i < a.size(); ++i) {
        diff += a[i] != b[i];
        sum += a[i];
    }
    printf("a 4 MB window of the two results: %llu differing counters, %llu tallies\n", (unsigned long long)diff, (unsigned long long)sum);
    return 0;
}
