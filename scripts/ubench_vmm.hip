// Is there a cheaper way to 100+ GB of device memory than hipMalloc's 28 ms per GB (round 5: what keeps the kept-lists
// route opt-in)?  After 60 GB held by plain hipMalloc (the product's state when the table stage starts), 8 segments of
// 16 GB each by
//   1. hipMalloc
//   2. hipMallocAsync from the device's default pool (release threshold = max), freed and taken again
//   3. hipMemAddressReserve + hipMemCreate + hipMemMap + hipMemSetAccess (2 MB granularity)
// ms per segment, then a kernel's first and second write of the segment.
//   hipcc --offload-arch=gfx950 -O2 scripts/ubench_vmm.hip -o scripts/bin/ubench_vmm && scripts/bin/ubench_vmm
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(uint32_t *p, size_t words)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}
static double touch_ms(void *p, size_t bytes)
{
    const double t0 = now();
    hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, 0, (uint32_t *)p, bytes / 4);
    (void)hipDeviceSynchronize();
    return (now() - t0) * 1e3;
}

int main()
{
    CK(hipSetDevice(0));
    const size_t seg = 16ull << 30;
    const int nseg = 8;
    std::vector<void *> held;
    for (int i = 0; i < 6; ++i) { // 60 GB held, touched
        void *p;
        CK(hipMalloc(&p, 10ull << 30));
        (void)touch_ms(p, 10ull << 30);
        held.push_back(p);
    }
    size_t fr, tot;
    CK(hipMemGetInfo(&fr, &tot));
    printf("holding 60 GB; free %.1f of %.1f GB\n", fr / 1e9, tot / 1e9);
    {   // 1. hipMalloc
        std::vector<void *> v;
        printf("hipMalloc, ms per 16 GB segment (+ first / second write):");
        for (int i = 0; i < nseg; ++i) {
            void *p;
            const double t0 = now();
            CK(hipMalloc(&p, seg));
            const double a = (now() - t0) * 1e3;
            const double w1 = touch_ms(p, seg), w2 = touch_ms(p, seg);
            printf(" %.0f (%.1f / %.1f)", a, w1, w2);
            v.push_back(p);
        }
        const double t0 = now();
        for (void *p : v) CK(hipFree(p));
        printf("\n  hipFree of all: %.0f ms\n", (now() - t0) * 1e3);
    }
    {   // 2. stream-ordered allocator, pool keeps what is freed
        hipMemPool_t pool;
        CK(hipDeviceGetDefaultMemPool(&pool, 0));
        uint64_t thr = ~0ull;
        CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr));
        for (int round = 0; round < 2; ++round) {
            std::vector<void *> v;
            printf("hipMallocAsync round %d, ms per segment (+ first write):", round);
            for (int i = 0; i < nseg; ++i) {
                void *p;
                const double t0 = now();
                CK(hipMallocAsync(&p, seg, 0));
                CK(hipStreamSynchronize(0));
                const double a = (now() - t0) * 1e3;
                printf(" %.0f (%.1f)", a, touch_ms(p, seg));
                v.push_back(p);
            }
            const double t0 = now();
            for (void *p : v) CK(hipFreeAsync(p, 0));
            CK(hipStreamSynchronize(0));
            printf("\n  hipFreeAsync of all: %.0f ms\n", (now() - t0) * 1e3);
        }
        CK(hipMemPoolTrimTo(pool, 0));
    }
    {   // 3. virtual memory management
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        size_t gran = 0;
        CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
        printf("VMM: recommended granularity %zu bytes\n", gran);
        void *base;
        CK(hipMemAddressReserve(&base, seg * nseg, 0, nullptr, 0));
        std::vector<hipMemGenericAllocationHandle_t> hs;
        printf("hipMemCreate + hipMemMap + hipMemSetAccess, ms per segment (create / map / access; first / second write):");
        for (int i = 0; i < nseg; ++i) {
            hipMemGenericAllocationHandle_t h;
            double t0 = now();
            CK(hipMemCreate(&h, seg, &prop, 0));
            const double c = (now() - t0) * 1e3;
            t0 = now();
            CK(hipMemMap((char *)base + seg * i, seg, 0, h, 0));
            const double m = (now() - t0) * 1e3;
            hipMemAccessDesc acc = {};
            acc.location = prop.location;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            t0 = now();
            CK(hipMemSetAccess((char *)base + seg * i, seg, &acc, 1));
            const double s = (now() - t0) * 1e3;
            const double w1 = touch_ms((char *)base + seg * i, seg), w2 = touch_ms((char *)base + seg * i, seg);
            printf(" %.0f/%.0f/%.0f (%.1f / %.1f)", c, m, s, w1, w2);
            hs.push_back(h);
        }
        const double t0 = now();
        for (int i = 0; i < nseg; ++i) {
            CK(hipMemUnmap((char *)base + seg * i, seg));
            CK(hipMemRelease(hs[i]));
        }
        CK(hipMemAddressFree(base, seg * nseg));
        printf("\n  unmap + release of all: %.0f ms\n", (now() - t0) * 1e3);
    }
    return 0;
}
