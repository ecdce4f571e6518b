// xcd_sync_probe.hip -- can the workgroups of ONE XCD (32 CUs, one L2) synchronise and exchange data through that L2
// alone?  Plain stores stay in the XCD's L2 (write-back), loads that bypass the CU's L1 (sc1) are served by it: so a
// barrier of per-workgroup flag words (one 128-byte line) and slabs of partial sums need no device-scope atomics,
// no release / acquire fences and no trip to the memory side -- IF every participant sits on the same XCD, which is
// checked at run time (HW_REG_XCC_ID), never assumed.  The probe launches one workgroup per CU, lets the ones on the
// elected XCD run R rounds of {write a slab, barrier, read everybody's slab and check it}, and reports the time per
// round and the number of stale reads.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/xcd_sync_probe.hip -o gpurun_out/xcd_sync_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t xcc_id()
{
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}

struct ctl_t {
    uint32_t arrivals[16];   // workgroups seen per XCC (memory-side atomics, once per launch)
    uint32_t pad[16];
    uint32_t flags[64];      // barrier words of the participants (rank-indexed)
    uint32_t bad, rounds_done, participants, elected;
};

template <int SLAB>   // floats per workgroup and round
__global__ __launch_bounds__(256) void probe_kernel(ctl_t *ctl, float *slabs, int rounds, uint64_t *t_out, int want_xcc)
{
    extern __shared__ uint32_t lds[];
    __shared__ uint32_t s_rank, s_go;
    const uint32_t me = xcc_id();
    if (threadIdx.x == 0) {
        s_rank = atomicAdd(&ctl->arrivals[me], 1u);
        s_go = (int)me == want_xcc ? 1u : 0u;
    }
    __syncthreads();
    if (!s_go) return;
    const uint32_t rank = s_rank;
    // how many take part: wait until the launch has placed everybody (all gridDim.x workgroups have arrived somewhere)
    uint32_t P = 0;
    if (threadIdx.x == 0) {
        for (int spin = 0; spin < (1 << 22); ++spin) {
            uint32_t tot = 0;
            for (int x = 0; x < 8; ++x) tot += __hip_atomic_load(&ctl->arrivals[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tot == gridDim.x) break;
            __builtin_amdgcn_s_sleep(2);
        }
        lds[0] = __hip_atomic_load(&ctl->arrivals[me], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    P = lds[0];
    __syncthreads();
    uint32_t bad = 0;
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 1; r <= rounds; ++r) {
        // my slab of this round: plain stores
        float *mine = slabs + (size_t)rank * SLAB;
        for (int i = threadIdx.x; i < SLAB; i += 256) mine[i] = (float)(r * 1000 + (int)rank) + (float)i * 0.001f;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_store(&ctl->flags[rank], (uint32_t)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // plain store
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // wait for everybody: lane l of wave 0 polls flag l (one line), L1 bypassed
        if (threadIdx.x < 64) {
            for (int spin = 0; spin < (1 << 22); ++spin) {
                const uint32_t f = threadIdx.x < P ? __hip_atomic_load(&ctl->flags[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (uint32_t)r;
                if (__all((int)(f >= (uint32_t)r))) break;
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        // everybody's slab, L1 bypassed: thread t sums element t of every slab (as the BatchNorm sums would be)
        for (int i = threadIdx.x; i < SLAB; i += 256) {
            for (uint32_t w = 0; w < P; ++w) {
                const float v = __hip_atomic_load(slabs + (size_t)w * SLAB + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float want = (float)(r * 1000 + (int)w) + (float)i * 0.001f;
                if (v != want) ++bad;
            }
        }
        // (a second barrier so that nobody overwrites a slab somebody is still reading)
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_store(&ctl->flags[32 + rank], (uint32_t)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (threadIdx.x < 64) {
            for (int spin = 0; spin < (1 << 22); ++spin) {
                const uint32_t f = threadIdx.x < P ? __hip_atomic_load(&ctl->flags[32 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (uint32_t)r;
                if (__all((int)(f >= (uint32_t)r))) break;
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
    const uint64_t t1 = __builtin_amdgcn_s_memrealtime();
    if (bad) atomicAdd(&ctl->bad, bad);
    if (threadIdx.x == 0) {
        t_out[rank] = t1 - t0;
        if (rank == 0) { ctl->participants = P; ctl->elected = me; ctl->rounds_done = rounds; }
    }
}

// barrier alone: MODE 0 flag words + sc1 polls with s_sleep, 1 the same without the sleep, 2 one device-scope counter
// (atomic add + sc1 poll: the form a kernel that spans XCDs needs)
template <int MODE>
__global__ __launch_bounds__(256) void barrier_kernel(ctl_t *ctl, int rounds, uint64_t *t_out, int want_xcc)
{
    extern __shared__ uint32_t lds[];
    __shared__ uint32_t s_rank, s_go;
    const uint32_t me = xcc_id();
    if (threadIdx.x == 0) {
        s_rank = atomicAdd(&ctl->arrivals[me], 1u);
        s_go = (int)me == want_xcc ? 1u : 0u;
    }
    __syncthreads();
    if (!s_go) return;
    const uint32_t rank = s_rank;
    if (threadIdx.x == 0) {
        for (int spin = 0; spin < (1 << 22); ++spin) {
            uint32_t tot = 0;
            for (int x = 0; x < 8; ++x) tot += __hip_atomic_load(&ctl->arrivals[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tot == gridDim.x) break;
            __builtin_amdgcn_s_sleep(2);
        }
        lds[0] = __hip_atomic_load(&ctl->arrivals[me], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const uint32_t P = lds[0];
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 1; r <= rounds; ++r) {
        __syncthreads();
        if (MODE == 2) {
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(&ctl->pad[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int spin = 0; spin < (1 << 22); ++spin) {
                    if (__hip_atomic_load(&ctl->pad[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (uint32_t)r * P) break;
                }
            }
        } else {
            if (threadIdx.x == 0) {
                __hip_atomic_store(&ctl->flags[rank], (uint32_t)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (threadIdx.x < 64) {
                for (int spin = 0; spin < (1 << 22); ++spin) {
                    const uint32_t f = threadIdx.x < P ? __hip_atomic_load(&ctl->flags[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (uint32_t)r;
                    if (__all((int)(f >= (uint32_t)r))) break;
                    if (MODE == 0) __builtin_amdgcn_s_sleep(1);
                }
            }
        }
        __syncthreads();
    }
    const uint64_t t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        t_out[rank] = t1 - t0;
        if (rank == 0) { ctl->participants = P; ctl->elected = me; ctl->rounds_done = rounds; }
    }
}

// float atomics at WORKGROUP scope on global memory: do they execute in the XCD's L2 (and so add up across the CUs of one
// XCD), and what does a round of {every workgroup adds 256 floats into the same 256 words, barrier, every workgroup reads the
// 256 sums with L1-bypassing loads, barrier} cost?  (agent-scope float atomics execute at the memory side: MI355X_MICROARCH.md)
template <int SCOPE>   // 0 workgroup scope, 1 agent scope
__global__ __launch_bounds__(256) void atomic_kernel(ctl_t *ctl, float *acc, int rounds, uint64_t *t_out, int want_xcc)
{
    extern __shared__ uint32_t lds[];
    __shared__ uint32_t s_rank, s_go;
    const uint32_t me = xcc_id();
    if (threadIdx.x == 0) {
        s_rank = atomicAdd(&ctl->arrivals[me], 1u);
        s_go = (int)me == want_xcc ? 1u : 0u;
    }
    __syncthreads();
    if (!s_go) return;
    const uint32_t rank = s_rank;
    if (threadIdx.x == 0) {
        for (int spin = 0; spin < (1 << 22); ++spin) {
            uint32_t tot = 0;
            for (int x = 0; x < 8; ++x) tot += __hip_atomic_load(&ctl->arrivals[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tot == gridDim.x) break;
            __builtin_amdgcn_s_sleep(2);
        }
        lds[0] = __hip_atomic_load(&ctl->arrivals[me], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const uint32_t P = lds[0];
    __syncthreads();
    auto barrier = [&](uint32_t *flags, int r) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_store(&flags[rank], (uint32_t)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (threadIdx.x < 64) {
            for (int spin = 0; spin < (1 << 22); ++spin) {
                const uint32_t f = threadIdx.x < P ? __hip_atomic_load(&flags[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (uint32_t)r;
                if (__all((int)(f >= (uint32_t)r))) break;
            }
        }
        __syncthreads();
    };
    uint32_t bad = 0;
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 1; r <= rounds; ++r) {
        float *a = acc + (r & 1) * 256;          // two buffers by round parity; the other one is cleared for the next round
        float *other = acc + ((r + 1) & 1) * 256;
        const float v = (float)(rank + 1);
        if (SCOPE == 0) __hip_atomic_fetch_add(&a[threadIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_fetch_add(&a[threadIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        barrier(ctl->flags, r);
        const float got = __hip_atomic_load(&a[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (got != (float)(P * (P + 1) / 2)) ++bad;
        if (rank == 0) other[threadIdx.x] = 0.0f;  // plain store: stays in the L2
        barrier(ctl->flags + 32, r);
    }
    const uint64_t t1 = __builtin_amdgcn_s_memrealtime();
    if (bad) atomicAdd(&ctl->bad, bad);
    if (threadIdx.x == 0) {
        t_out[rank] = t1 - t0;
        if (rank == 0) { ctl->participants = P; ctl->elected = me; ctl->rounds_done = rounds; }
    }
}

template <int SCOPE> static int run_atomic(ctl_t *ctl, float *acc, uint64_t *t, int cus, const char *what)
{
    CK(hipFuncSetAttribute((const void *)atomic_kernel<SCOPE>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(ctl, 0, sizeof(ctl_t))); CK(hipMemset(t, 0, 64 * 8)); CK(hipMemset(acc, 0, 512 * 4));
        const int rounds = 2000;
        hipLaunchKernelGGL(atomic_kernel<SCOPE>, dim3(cus), dim3(256), 96 * 1024, 0, ctl, acc, rounds, t, rep);
        CK(hipDeviceSynchronize());
        ctl_t h; CK(hipMemcpy(&h, ctl, sizeof h, hipMemcpyDeviceToHost));
        std::vector<uint64_t> ht(64); CK(hipMemcpy(ht.data(), t, 64 * 8, hipMemcpyDeviceToHost));
        printf("float atomics, %s: XCC %u, %u participants, wrong sums %u of %d, %.3f us per round {256 adds per workgroup, barrier, read, barrier}\n",
               what, h.elected, h.participants, h.bad, rounds * 256 * (int)h.participants, ht[0] / 100.0 / rounds);
    }
    return 0;
}

template <int MODE> static int run_barrier(ctl_t *ctl, uint64_t *t, int cus, const char *what)
{
    CK(hipFuncSetAttribute((const void *)barrier_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(ctl, 0, sizeof(ctl_t))); CK(hipMemset(t, 0, 64 * 8));
        const int rounds = 5000;
        hipLaunchKernelGGL(barrier_kernel<MODE>, dim3(cus), dim3(256), 96 * 1024, 0, ctl, rounds, t, rep);
        CK(hipDeviceSynchronize());
        ctl_t h; CK(hipMemcpy(&h, ctl, sizeof h, hipMemcpyDeviceToHost));
        std::vector<uint64_t> ht(64); CK(hipMemcpy(ht.data(), t, 64 * 8, hipMemcpyDeviceToHost));
        printf("barrier alone, %s: XCC %u, %u participants, %.3f us per barrier\n", what, h.elected, h.participants, ht[0] / 100.0 / rounds);
    }
    return 0;
}

int main()
{
    ctl_t *ctl; float *slabs; uint64_t *t;
    CK(hipMalloc(&ctl, sizeof(ctl_t))); CK(hipMalloc(&slabs, 64 * 4096 * 4)); CK(hipMalloc(&t, 64 * 8));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    CK(hipFuncSetAttribute((const void *)probe_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    CK(hipFuncSetAttribute((const void *)probe_kernel<4096>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    if (run_barrier<0>(ctl, t, cus, "flag words in the XCD's L2, polls with s_sleep 1")) return 1;
    if (run_barrier<1>(ctl, t, cus, "flag words in the XCD's L2, polls back to back")) return 1;
    if (run_barrier<2>(ctl, t, cus, "one device-scope counter (atomic add + poll)")) return 1;
    if (run_atomic<0>(ctl, slabs, t, cus, "workgroup scope")) return 1;
    if (run_atomic<1>(ctl, slabs, t, cus, "agent scope")) return 1;
    for (int slab : {256}) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(ctl, 0, sizeof(ctl_t))); CK(hipMemset(t, 0, 64 * 8));
            const int rounds = 2000;
            // 96 KB of LDS per workgroup: one workgroup per CU
            if (slab == 256) hipLaunchKernelGGL(probe_kernel<256>, dim3(cus), dim3(256), 96 * 1024, 0, ctl, slabs, rounds, t, rep % 8);
            else hipLaunchKernelGGL(probe_kernel<4096>, dim3(cus), dim3(256), 96 * 1024, 0, ctl, slabs, rounds, t, rep % 8);
            CK(hipDeviceSynchronize());
            ctl_t h; CK(hipMemcpy(&h, ctl, sizeof h, hipMemcpyDeviceToHost));
            std::vector<uint64_t> ht(64); CK(hipMemcpy(ht.data(), t, 64 * 8, hipMemcpyDeviceToHost));
            printf("slab %5d floats: XCC %u, %u participants (arrivals per XCC:", slab, h.elected, h.participants);
            for (int x = 0; x < 8; ++x) printf(" %u", h.arrivals[x]);
            printf("), %u rounds, stale reads %u, %.3f us per round {write, barrier, read %u slabs, barrier}\n", h.rounds_done, h.bad,
                   ht[0] / 100.0 / rounds, h.participants);
        }
    }
    return 0;
}
