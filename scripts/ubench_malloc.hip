// What a large hipMalloc costs, and whom it holds up (round 5: can the slice-list pool and the big workspaces be
// allocated on a side thread while the composition stage parses and launches?).
//   hipcc --offload-arch=gfx950 -O2 scripts/ubench_malloc.hip -o /tmp/ubench_malloc -lpthread && /tmp/ubench_malloc
// 1. hipMalloc / first-touch kernel / hipFree of 1, 4, 16, 64 GB, twice each (is the second one cheaper?)
// 2. a 64 GB hipMalloc on a side thread while the main thread (a) launches + synchronises a short kernel in a loop,
//    (b) does 1 MB hipMalloc / hipFree pairs, (c) hipMemcpyAsync H2D of 64 MB from pinned memory: the slowest and the
//    mean iteration of each loop with and without the side thread
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

#define CK(x)                                                                        \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                           \
            return 1;                                                                \
        }                                                                            \
    } while (0)

static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

__global__ void touch(uint32_t *p, size_t words)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}

__global__ void tiny(uint32_t *p) { p[threadIdx.x] += 1; }

struct loop_stat {
    double mean_us, max_us;
    int iters;
};

template <typename F>
static loop_stat run_loop(std::atomic<bool> &stop, double min_s, F body)
{
    loop_stat s{0, 0, 0};
    const double t0 = now();
    double sum = 0;
    while (!stop.load() || now() - t0 < min_s) {
        const double a = now();
        body();
        const double d = (now() - a) * 1e6;
        sum += d;
        if (d > s.max_us) s.max_us = d;
        ++s.iters;
        if (now() - t0 > 20) break;
    }
    s.mean_us = sum / (s.iters ? s.iters : 1);
    return s;
}

int main()
{
    CK(hipSetDevice(0));
    size_t fr, tot;
    CK(hipMemGetInfo(&fr, &tot));
    printf("free %.1f GB of %.1f GB\n", fr / 1e9, tot / 1e9);
    for (size_t gb : {1, 4, 16, 64}) {
        for (int rep = 0; rep < 2; ++rep) {
            void *p;
            double t0 = now();
            CK(hipMalloc(&p, gb << 30));
            double t1 = now();
            hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, 0, (uint32_t *)p, (gb << 30) / 4);
            CK(hipDeviceSynchronize());
            double t2 = now();
            hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, 0, (uint32_t *)p, (gb << 30) / 4);
            CK(hipDeviceSynchronize());
            double t3 = now();
            CK(hipFree(p));
            double t4 = now();
            printf("%3zu GB: hipMalloc %8.1f ms (%6.1f GB/s)  first write %7.1f ms  second write %7.1f ms  hipFree %7.1f ms\n", gb,
                   (t1 - t0) * 1e3, gb * 1.0737 / (t1 - t0), (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3);
        }
    }
    // ---- where does the cost start?  segments held, one after the other ----
    for (size_t seg_gb : {8, 1}) {
        std::vector<void *> held;
        const int count = seg_gb == 8 ? 30 : 120;
        printf("%d x %zu GB held, ms each:", count, seg_gb);
        for (int i = 0; i < count; ++i) {
            void *p = nullptr;
            const double t0 = now();
            if (hipMalloc(&p, seg_gb << 30) != hipSuccess) {
                (void)hipGetLastError();
                printf(" (out of memory at %d)", i);
                break;
            }
            printf(" %.0f", (now() - t0) * 1e3);
            held.push_back(p);
        }
        const double t0 = now();
        for (void *p : held) (void)hipFree(p);
        printf("\n   freed in %.0f ms\n", (now() - t0) * 1e3);
    }
    // ---- who waits for a big allocation on another thread? ----
    uint32_t *d_small;
    CK(hipMalloc((void **)&d_small, 4096));
    CK(hipMemset(d_small, 0, 4096));
    void *h_pin, *d_dst;
    CK(hipHostMalloc(&h_pin, 64 << 20));
    CK(hipMalloc(&d_dst, 64 << 20));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    for (int with_side = 0; with_side < 2; ++with_side) {
        for (int what = 0; what < 3; ++what) {
            std::atomic<bool> stop{with_side == 0};
            double side_ms = 0;
            std::thread side;
            if (with_side)
                side = std::thread([&] {
                    (void)hipSetDevice(0);
                    void *p[4] = {};
                    const double t0 = now();
                    for (int i = 0; i < 4; ++i) (void)hipMalloc(&p[i], 16ull << 30); // four 16 GB segments, as the pool would
                    side_ms = (now() - t0) * 1e3;
                    stop.store(true);
                    for (int i = 0; i < 4; ++i) (void)hipFree(p[i]);
                });
            loop_stat s;
            if (what == 0)
                s = run_loop(stop, 0.5, [&] {
                    hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, d_small);
                    (void)hipStreamSynchronize(st);
                });
            else if (what == 1)
                s = run_loop(stop, 0.5, [&] {
                    void *q;
                    (void)hipMalloc(&q, 1 << 20);
                    (void)hipFree(q);
                });
            else
                s = run_loop(stop, 0.5, [&] {
                    (void)hipMemcpyAsync(d_dst, h_pin, 64 << 20, hipMemcpyHostToDevice, st);
                    (void)hipStreamSynchronize(st);
                });
            if (with_side) side.join();
            printf("%s | %-28s: %6d iterations, mean %9.1f us, slowest %9.1f us%s\n", with_side ? "4 x 16 GB hipMalloc on a side thread" : "alone                              ",
                   what == 0 ? "tiny kernel + sync" : what == 1 ? "1 MB hipMalloc + hipFree" : "64 MB H2D from pinned + sync", s.iters, s.mean_us, s.max_us,
                   with_side ? "" : "");
            if (with_side) printf("      (the side thread's four allocations took %.1f ms)\n", side_ms);
        }
    }
    return 0;
}
