// overlap_probe.hip -- does a chain of DEPENDENT short kernels get shorter when consecutive kernels sit on two
// streams and the dependence is a flag in memory instead of stream order?
//
// The VAE step is eleven launches of 5-13 us, each waiting for the one before (BatchNorm needs the whole batch).  A
// launch costs its dispatch, the kernel-argument loads and the wave start before the first useful load is issued,
// and the end-of-kernel drain after the last store; in stream order none of that overlaps with the neighbour.  Here
// kernel i+1 is launched on the OTHER stream: it starts while kernel i still runs, does whatever does not depend on
// kernel i (here: nothing, or a 64 KB read that stands for its weights), then waits until `done[i]` counts all of
// kernel i's workgroups (release / acquire at agent scope: the producer's L2 is written back, the consumer's
// invalidated), then does its dependent part.  Kernel i+2 is stream-ordered behind kernel i, so never more than two
// kernels are in flight and every waiting workgroup has a running producer: no deadlock as long as two launches fit
// on the chip together (128 workgroups of < 80 KB LDS each: they do).  Every wait has a time-out that sets an error
// word instead of hanging.
//
// Checked, not assumed: each kernel reads what OTHER workgroups of its predecessor wrote and adds one; after n
// kernels every element must be n.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/overlap_probe.hip -o gpurun_out/overlap_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int WG = 128, TPB = 256, PER = 16;          // 128 workgroups x 256 threads x 16 floats = 2 MB per buffer
constexpr int N_EL = WG * TPB * PER;

struct args_t {
    const float *in;
    float *out;
    const float *weights;    // 64 KB read by every workgroup before the wait (stands for the layer's weights)
    uint32_t *done;          // done[i]: workgroups of kernel i that have finished
    uint32_t *err;
    uint32_t idx;            // global index of this kernel in the chain
    int use_flags, work, prologue, lds_bytes;
};

__global__ __launch_bounds__(TPB) void link_kernel(args_t a)
{
    extern __shared__ float lds[];
    const int tid = threadIdx.x, wg = blockIdx.x;
    float pro = 0.0f;
    if (a.prologue) {
        // independent of the predecessor: 64 KB into LDS
        for (int i = tid; i < 16384; i += TPB) lds[i] = a.weights[i];
        __syncthreads();
        pro = lds[(tid * 61) & 16383];
    }
    if (a.use_flags && a.idx > 0) {
        if (tid == 0) {
            const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
            // relaxed polls (one L2-bypassing load each); ONE acquire once the count is there
            while (__hip_atomic_load(&a.done[a.idx - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)WG) {
                __builtin_amdgcn_s_sleep(1);
                if (__builtin_amdgcn_s_memrealtime() - t0 > 100u * 1000u * 2u) {   // 2 ms at 100 MHz
                    atomicOr(a.err, 1u);
                    break;
                }
            }
            if (a.use_flags == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    // dependent part: what workgroup (wg * 37 + 11) % WG of the predecessor wrote
    const int src = (wg * 37 + 11) % WG;
    const float4 *p = reinterpret_cast<const float4 *>(a.in + ((size_t)src * TPB + tid) * PER);
    float4 v[PER / 4];
#pragma unroll
    for (int j = 0; j < PER / 4; ++j) v[j] = p[j];
    float f = pro * 0.0f;
    for (int w = 0; w < a.work; ++w) f = fmaf(f, 1.0000001f, v[w & 3].x * 1e-30f);
    float4 *q = reinterpret_cast<float4 *>(a.out + ((size_t)wg * TPB + tid) * PER);
#pragma unroll
    for (int j = 0; j < PER / 4; ++j) q[j] = make_float4(v[j].x + 1.0f + f * 0.0f, v[j].y + 1.0f, v[j].z + 1.0f, v[j].w + 1.0f);
    if (a.use_flags) {
        // every thread's stores have left the CU, then ONE release (the write-back of this XCD's L2) and the count
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (tid == 0) {
            if (a.use_flags == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(&a.done[a.idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

int main(int argc, char **argv)
{
    const int chain = argc > 1 ? atoi(argv[1]) : 110;   // kernels per graph (ten steps of eleven)
    const int reps = argc > 2 ? atoi(argv[2]) : 30;
    float *buf[2], *weights;
    uint32_t *done, *err;
    CK(hipMalloc(&buf[0], N_EL * 4)); CK(hipMalloc(&buf[1], N_EL * 4)); CK(hipMalloc(&weights, 65536));
    CK(hipMalloc(&done, sizeof(uint32_t) * (chain + 1))); CK(hipMalloc(&err, 4));
    CK(hipMemset(weights, 0, 65536));
    hipStream_t s[2];
    CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
    hipEvent_t e0, e1, fork, join;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    CK(hipFuncSetAttribute((const void *)link_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    std::vector<float> host(N_EL);
    printf("chain of %d dependent kernels (128 workgroups x 256 threads, 2 MB in, 2 MB out), %d runs of the chain\n", chain, reps);
    printf("%-66s %10s %8s %8s\n", "form", "us/kernel", "us/11", "check");
    // flags 0: stream order only   1: count + release / acquire fences (one thread per workgroup)   2: count only
    struct cfg { int graph, two, flags, prologue, work; };
    std::vector<cfg> cfgs;
    for (int work : {0, 300})
        for (int graph : {1, 0})
            for (int two : {0, 1})
                for (int flags : {0, 1, 2})
                    for (int prologue : {0, 1}) {
                        if (two && !flags) continue;   // two streams need the flag
                        cfgs.push_back({graph, two, flags, prologue, work});
                    }
    for (const cfg &c : cfgs) {
        const int lds_bytes = c.prologue ? 65536 : 0;
        auto enqueue = [&]() -> int {
            CK(hipMemsetAsync(done, 0, sizeof(uint32_t) * (chain + 1), s[0]));
            if (c.two) { CK(hipEventRecord(fork, s[0])); CK(hipStreamWaitEvent(s[1], fork, 0)); }
            for (int i = 0; i < chain; ++i) {
                args_t a{buf[i & 1], buf[(i + 1) & 1], weights, done, err, (uint32_t)i, c.flags, c.work, c.prologue, lds_bytes};
                hipLaunchKernelGGL(link_kernel, dim3(WG), dim3(TPB), lds_bytes, s[c.two ? (i & 1) : 0], a);
            }
            if (c.two) { CK(hipEventRecord(join, s[1])); CK(hipStreamWaitEvent(s[0], join, 0)); }
            return 0;
        };
        hipGraph_t g = nullptr;
        hipGraphExec_t ge = nullptr;
        if (c.graph) {
            CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal));
            if (enqueue()) return 1;
            CK(hipStreamEndCapture(s[0], &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        }
        auto run = [&]() -> int {
            if (c.graph) CK(hipGraphLaunch(ge, s[0]));
            else if (enqueue()) return 1;
            return 0;
        };
        CK(hipMemset(err, 0, 4));
        CK(hipMemset(buf[0], 0, N_EL * 4));
        if (run()) return 1;
        CK(hipStreamSynchronize(s[0]));
        CK(hipMemcpy(host.data(), buf[chain & 1], N_EL * 4, hipMemcpyDeviceToHost));
        long bad = 0;
        for (int i = 0; i < N_EL; ++i) bad += host[i] != (float)chain;
        uint32_t herr;
        CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        const int nrep = herr ? 1 : reps;   // (a form that timed out once is not worth thirty more runs)
        CK(hipEventRecord(e0, s[0]));
        for (int r = 0; r < nrep; ++r)
            if (run()) return 1;
        CK(hipEventRecord(e1, s[0]));
        CK(hipStreamSynchronize(s[0]));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        char name[128];
        snprintf(name, sizeof name, "%s, %s%s%s, work %d", c.graph ? "graph" : "launches", c.two ? "two streams" : "one stream",
                 c.flags == 1 ? " + count + fences" : c.flags == 2 ? " + count" : "", c.prologue ? " + 64 KB prologue" : "", c.work);
        const double us = ms * 1e3 / ((double)nrep * chain);
        printf("%-66s %10.2f %8.1f %8s%s\n", name, us, us * 11, bad ? "STALE" : "ok", herr ? "  (time-outs!)" : "");
        fflush(stdout);
        if (c.graph) { CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); }
    }
    return 0;
}
