// k4s2_probe.hip -- where a workgroup of the stride-2 k = 4 kernel (k1_lane4s2_kernel) spends its time: launches the
// kernel on 1 M synthetic 10 kb reads with one half group per workgroup and with resident workgroups walking the half
// groups, times both, and prints the s_memrealtime stamps of a few workgroups (tally / flush per half group).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ilrbinner_amd/csrc scripts/k4s2_probe.hip \
//         -Llrbinner_amd -llrb_hip -Wl,-rpath,$PWD/lrbinner_amd -o gpurun_out/k4s2_probe
#include "../lrbinner_amd/csrc/lrb_kernels.hip"
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int W, int NR> static int run(const char *name, unsigned grid, const uint4 *ct, const uint64_t *goff, const uint32_t *lens,
                                        uint64_t n, uint32_t *counts, uint64_t *dbg, bool print)
{
    constexpr size_t smem = 65536 + 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k1_lane4s2_kernel<W, NR, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k1_lane4s2_kernel<W, NR, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 40; ++i) hipLaunchKernelGGL((k1_lane4s2_kernel<W, NR, false>), dim3(grid), dim3(64 * W), smem, 0, ct, goff, nullptr, lens, n, counts, nullptr);
    CK(hipEventRecord(a));
    const int reps = 50;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k1_lane4s2_kernel<W, NR, false>), dim3(grid), dim3(64 * W), smem, 0, ct, goff, nullptr, lens, n, counts, nullptr);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-28s grid %6u  %.3f ms  roofline %.3f\n", name, grid, ms / reps, 3044.0 * n / (ms / reps * 1e-3) / 8e12);
    if (!print) return 0;
    CK(hipMemset(dbg, 0, (size_t)grid * 256 * 8));
    hipLaunchKernelGGL((k1_lane4s2_kernel<W, NR, true>), dim3(grid), dim3(64 * W), smem, 0, ct, goff, nullptr, lens, n, counts, dbg);
    CK(hipDeviceSynchronize());
    std::vector<uint64_t> h((size_t)grid * 256);
    CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
    uint64_t t0 = ~0ull, t1 = 0;
    for (unsigned g = 0; g < grid; ++g) { t0 = std::min(t0, h[(size_t)g * 256 + 1]); t1 = std::max(t1, h[(size_t)g * 256 + 1 + h[(size_t)g * 256]]); }
    printf("  stamped launch: %.1f us first start -> last end\n", (t1 - t0) / 100.0);
    // per half group: tally = s[3i+1]-s[3i], flush = s[3i+2]-s[3i+1], gap to the next = s[3i+3]-s[3i+2]
    double st = 0, sf = 0, sg = 0; uint64_t cnt = 0, cg = 0;
    for (unsigned g = 0; g < grid; ++g) {
        const uint64_t *s = &h[(size_t)g * 256 + 1];
        const uint64_t ns = h[(size_t)g * 256];
        for (uint64_t i = 0; i + 2 < ns; i += 3) {
            st += s[i + 1] - s[i]; sf += s[i + 2] - s[i + 1]; ++cnt;
            if (i + 3 < ns) { sg += s[i + 3] - s[i + 2]; ++cg; }
        }
    }
    printf("  mean per half group: tally %.2f us, flush %.2f us, flush end -> next tally start %.2f us (%llu groups)\n",
           st / cnt / 100, sf / cnt / 100, cg ? sg / cg / 100 : 0.0, (unsigned long long)cnt);
    for (unsigned g : {0u, 1u, grid / 2, grid - 1}) {
        const uint64_t *s = &h[(size_t)g * 256 + 1];
        printf("  wg %u:", g);
        for (uint64_t i = 0; i < std::min<uint64_t>(h[(size_t)g * 256] + 1, 13); ++i) printf(" %.1f", (s[i] - t0) / 100.0);
        printf("\n");
    }
    return 0;
}

int main()
{
    const uint64_t n = 1000000, L = 10000;
    const uint64_t ngroups = (n + 63) / 64, rows = (L + 63) / 64 + 1;
    std::vector<uint64_t> goff(ngroups + 1);
    for (uint64_t g = 0; g <= ngroups; ++g) goff[g] = g * rows;
    uint4 *ct; uint64_t *d_goff, *dbg; uint32_t *lens, *counts;
    const size_t ct_bytes = ngroups * rows * 1024;
    CK(hipMalloc(&ct, ct_bytes)); CK(hipMalloc(&d_goff, goff.size() * 8)); CK(hipMalloc(&lens, n * 4));
    CK(hipMalloc(&counts, n * 136 * 4)); CK(hipMalloc(&dbg, (size_t)2 * ngroups * 256 * 8));
    {
        std::vector<uint32_t> h(ct_bytes / 4);
        std::mt19937 rng(5);
        for (auto &x : h) x = rng();
        // halo row of every group is zero, as the layout says
        for (uint64_t g = 0; g < ngroups; ++g) std::fill(h.begin() + ((g + 1) * rows - 1) * 256, h.begin() + (g + 1) * rows * 256, 0u);
        CK(hipMemcpy(ct, h.data(), ct_bytes, hipMemcpyHostToDevice));
        std::vector<uint32_t> l(n, (uint32_t)L);
        CK(hipMemcpy(lens, l.data(), n * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_goff, goff.data(), goff.size() * 8, hipMemcpyHostToDevice));
    }
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const unsigned cus = prop.multiProcessorCount;
    printf("%s, %u CUs\n", prop.gcnArchName, cus);
    if (run<8, 2>("W8 NR2 one half group / wg", (unsigned)(2 * ngroups), ct, d_goff, lens, n, counts, dbg, true)) return 1;
    if (run<8, 2>("W8 NR2 resident x2", 2 * cus, ct, d_goff, lens, n, counts, dbg, true)) return 1;
    if (run<8, 4>("W8 NR4 one half group / wg", (unsigned)(2 * ngroups), ct, d_goff, lens, n, counts, dbg, false)) return 1;
    if (run<8, 4>("W8 NR4 resident x2", 2 * cus, ct, d_goff, lens, n, counts, dbg, false)) return 1;
    return 0;
}
