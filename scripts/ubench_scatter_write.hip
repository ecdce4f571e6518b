// ubench_scatter_write.hip -- what the part kernel's write pattern costs the memory system (gfx950).
// wl_part_kernel appends, per tile of 16,384 windows, one run of ~64 entries (256 bytes) to each of its unit's 256 slice
// lists: 2 x 256 workgroups x 256 streams, every stream advancing 256 bytes per 14 us.  This writes the same total (16 GB)
// in that shape -- W workgroups x S streams, a chunk of C bytes per stream and round, a wave's 16-byte stores covering
// 1,024 / C ... chunks -- for C = 256 B .. 4 KB, and as one plain stream, and prints TB/s.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_scatter_write.hip -o gpurun_out/ubench_scatter_write && gpurun_out/ubench_scatter_write
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

// stream (w, s) = bytes [(w S + s) len, + len); round r writes chunk r of every stream of the workgroup
__global__ __launch_bounds__(1024) void scatter(uint8_t *__restrict__ dst, uint32_t S, uint32_t C, uint32_t rounds, uint64_t len)
{
    const uint32_t tid = threadIdx.x;
    const uint32_t per_chunk = C / 16;                 // lanes (16-byte stores) a chunk
    const uint32_t chunks_per_pass = 1024 / per_chunk; // chunks the workgroup writes with one store a thread
    const v4u val = {tid, blockIdx.x, 3u, 4u};
    for (uint32_t r = 0; r < rounds; ++r) {
        for (uint32_t c0 = 0; c0 < S; c0 += chunks_per_pass) {
            const uint32_t s = c0 + tid / per_chunk;
            if (s < S) {
                uint8_t *p = dst + ((uint64_t)blockIdx.x * S + s) * len + (uint64_t)r * C + (tid % per_chunk) * 16;
                *reinterpret_cast<v4u *>(p) = val;
            }
        }
        __syncthreads();
    }
}

// the same with runs of E entries (4 bytes each; E not a multiple of 4: the run boundaries wander through the 16-byte and
// 128-byte grid) written as wl_copy_run does: single stores up to the 16-byte boundary, 16-byte stores, single stores
__global__ __launch_bounds__(1024) void scatter_runs(uint32_t *__restrict__ dst, uint32_t S, uint32_t E, uint32_t rounds, uint64_t len)
{
    const uint32_t tid = threadIdx.x, sub = tid & 15u;
    for (uint32_t r = 0; r < rounds; ++r) {
        for (uint32_t c0 = 0; c0 < S; c0 += 64) {
            const uint32_t s = c0 + (tid >> 4);
            uint32_t *d = dst + ((uint64_t)blockIdx.x * S + s) * len + (uint64_t)r * E;
            uint32_t a = (4u - ((uint32_t)((uintptr_t)d >> 2) & 3u)) & 3u;
            if (a > E) a = E;
            const uint32_t nvec = (E - a) >> 2, t0 = a + 4 * nvec;
            if (sub < a) d[sub] = tid;
            else if (sub >= 4 && sub - 4 < E - t0) d[t0 + sub - 4] = tid;
            for (uint32_t v = sub; v < nvec; v += 16) {
                const v4u x = {tid, r, 3u, 4u};
                *reinterpret_cast<v4u *>(d + a + 4 * v) = x;
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void stream(uint8_t *__restrict__ dst, uint64_t n16)
{
    const v4u val = {threadIdx.x, blockIdx.x, 3u, 4u};
    for (uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 1024)
        reinterpret_cast<v4u *>(dst)[i] = val;
}

// a plain read stream for comparison (16 bytes a lane, 4 loads in flight a lane)
__global__ __launch_bounds__(1024) void read_stream(const uint8_t *__restrict__ src, uint64_t n16, uint32_t *out)
{
    const v4u *p = reinterpret_cast<const v4u *>(src);
    uint32_t acc = 0;
    const uint64_t stride = (uint64_t)gridDim.x * 1024;
    for (uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x; i + 3 * stride < n16; i += 4 * stride) {
        const v4u a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc += a.x ^ b.y ^ c.z ^ d.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main()
{
    const uint64_t total = 16ull << 30;
    uint8_t *buf;
    CHECK(hipMalloc(&buf, total));
    CHECK(hipMemset(buf, 0, total));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float ms;
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(stream, dim3(2048), dim3(1024), 0, 0, buf, total / 16);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("plain stream            : %7.3f ms  %6.2f TB/s\n", ms, total / ms / 1e9);
    }
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(read_stream, dim3(2048), dim3(1024), 0, 0, buf, total / 16, (uint32_t *)buf);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("plain read stream       : %7.3f ms  %6.2f TB/s\n", ms, total / ms / 1e9);
    }
    const uint32_t W = 512, S = 256;
    const uint64_t len = total / W / S; // 128 KB a stream
    for (uint32_t C = 128; C <= 8192; C *= 2) {
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(scatter, dim3(W), dim3(1024), 0, 0, buf, S, C, (uint32_t)(len / C), len);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("chunks of %5u bytes    : %7.3f ms  %6.2f TB/s  (%u workgroups x %u streams, %u rounds)\n", C, ms, total / ms / 1e9, W, S,
                   (uint32_t)(len / C));
        }
    }
    for (uint32_t E : {64u, 63u, 48u, 80u, 16u, 32u, 40u, 72u, 8u}) {
        const uint64_t len32 = len / 4;
        const uint32_t rounds = (uint32_t)(len32 / E);
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(scatter_runs, dim3(W), dim3(1024), 0, 0, (uint32_t *)buf, S, E, rounds, len32);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double bytes = (double)W * S * rounds * E * 4;
            printf("runs of %4u entries     : %7.3f ms  %6.2f TB/s  (%u rounds)\n", E, ms, bytes / ms / 1e9, rounds);
        }
    }
    // the same with the streams of a workgroup 40 KB long (a unit's slice list) and 2 workgroups a CU in flight
    CHECK(hipGetLastError());
    return 0;
}
