// lrb_lists.hip -- K2 and K3 on ONE partition of the 15-mer windows (round 4 form; the part kernel in its round-5 form).
//
// Replaces line_to_kmer_counts (kmer_utils.h:114-156) and line_to_vec (kmer_utils.h:24-87) for large batches.
// The table is kept as its canonical half H[2^29] (pair index h of a window: lrb_k15_dev.h), the coverage map as
// one byte per pair.  A random gather or atomic that misses the L2 costs a 128-byte line each (55 G/s on the whole
// chip), so the windows are brought to the table instead of the table to the windows:
//
//   count   (wl_count_kernel)  the reads of a GROUP (<= 2048 reads, sized so that the groups fill the CUs in whole
//           rounds) are cut into 1-4 UNITS, a workgroup per unit: the unit's windows tallied by 2 MB map slice (top
//           8 bits of h) in lane-private LDS counters; a scan kernel then places every unit's run of every slice.
//   part    (wl_part_kernel)   a workgroup per unit appends the unit's windows to its 256 level-1 lists in a scratch
//           buffer through 256 rings of 128 entries in LDS (a slot = the list position mod 128, a ring's quarter = a
//           128-byte line of the list): ONE returning 64-bit atomic a window -- its position, and whether the ring has
//           room -- and one store; every 16 k windows the complete lines go out, four lanes a line.
//           Entry = {read in the group : 11 | offset in the slice : 21}.
//   order   (wl_order_kernel_occ1 / wl_order_kernel)  a workgroup takes the (group, slice) lists of its slice in turn
//           and orders each by the next 6 bits of h (64 BUCKETS of 2^15 pairs per slice) into the group's final list:
//           the list held in registers (the next one asked for meanwhile), tallied once, 16 k-entry tiles ranked and
//           copied out; lists of more than 65,536 entries streamed twice by the second kernel.
//           bounds[g][b] = where bucket b starts in group g's region.
//   tally   (wl_tally_kernel)  K2: a workgroup per bucket walks every group's run of that bucket with the bucket's
//           2^15 counters in LDS and adds them to H as one coalesced read-modify-write.
//   sweep   (wl_sweep_kernel)  K3: a workgroup per group keeps the group's histograms in LDS ([bin][read] u16) and
//           walks the 16,384 buckets in order; the bucket's 32 KB of the map are STAGED IN LDS (registers loaded one
//           step ahead, list entries eight steps ahead), so a window costs one LDS byte read and one LDS add -- no
//           L2 request per window.  The workgroups of an XCD walk the map in step: each 32 KB leaves HBM once per
//           XCD and round.
//
// 4 B written + 4 B read (part -> order), 4 B written (order), 4 B read (tally), 4 B read (sweep) per window.
// Reads of more than 65,535 windows (a u16 counter could overflow) are left out here; the callers tally them with
// the gather / atomic kernels of lrb_kernels.hip.  Same H and same histograms as those kernels, bit for bit.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <vector>

#include "lrb_device.h"
#include "lrb_k15_dev.h"

#define WL_SLICE_BITS 21u
#define WL_SLICES 256u
#define WL_OFF_MASK ((1u << WL_SLICE_BITS) - 1u)
#define WL_SUB_BITS 15u           // a bucket = 2^15 pairs = 32 KB of the map
#define WL_SUBS 64u               // buckets per slice
#define WL_BUCKETS 16384u         // 2^29 >> 15
#define WL_BSTRIDE (WL_BUCKETS + 1u)
#define WL_TILE 16384u
#define WL_MAX_READS 2048u
#define WL_MAX_WINDOWS 65535u
#define WL_MAX_UNITS 4u
#define WL_HIST_CAP 65024u        // u16 counters of a group's histograms (reads rounded up to even): 127 KB beside the 32 KB map bucket

typedef uint32_t wl_v4u __attribute__((ext_vector_type(4)));
// Order this wave's LDS traffic across lanes.  DS operations of one wave execute in issue order, so only the compiler
// has to be kept from moving them.
__device__ __forceinline__ void wl_wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Prefix sums across lanes by DPP (VALU data paths) instead of __shfl_up (ds_bpermute: an LDS round trip per step, six
// steps deep in every scan below, on kernels whose LDS queue is busy with atomics).  Lanes without a source add 0.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ uint32_t wl_dpp(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
// inclusive scan over the 64 lanes: row_shr 1, 2, 4, 8 inside the rows of 16, then the rows' totals (row_bcast 15 / 31)
__device__ __forceinline__ uint32_t wl_wave_scan_incl(uint32_t v)
{
    v += wl_dpp<0x111>(v);
    v += wl_dpp<0x112>(v);
    v += wl_dpp<0x114>(v);
    v += wl_dpp<0x118>(v);
    v += wl_dpp<0x142, 0xA>(v);
    v += wl_dpp<0x143, 0xC>(v);
    return v;
}
// inclusive scan inside aligned groups of W lanes (W = 4, 8 or 16): a row shift never leaves its row of 16, the guard
// keeps it inside the group
template <int W>
__device__ __forceinline__ uint32_t wl_group_scan_incl(uint32_t v, uint32_t lane)
{
    static_assert(W == 4 || W == 8 || W == 16, "groups inside a DPP row");
    const uint32_t l = lane & (W - 1);
    uint32_t up = wl_dpp<0x111>(v);
    if (l >= 1) v += up;
    up = wl_dpp<0x112>(v);
    if (l >= 2) v += up;
    if (W > 4) {
        up = wl_dpp<0x114>(v);
        if (l >= 4) v += up;
    }
    if (W > 8) {
        up = wl_dpp<0x118>(v);
        if (l >= 8) v += up;
    }
    return v;
}



// exclusive scan of 256 values by ONE wave (lane l owns four consecutive ones); returns the total in every lane
template <typename LoadF, typename StoreF>
__device__ __forceinline__ uint32_t wl_wave_scan256(uint32_t lane, LoadF load, StoreF store)
{
    uint32_t v[4], own = 0;
#pragma unroll
    for (uint32_t q = 0; q < 4; ++q) {
        v[q] = load(lane * 4 + q);
        own += v[q];
    }
    const uint32_t inc = wl_wave_scan_incl(own);
    uint32_t run = inc - own;
#pragma unroll
    for (uint32_t q = 0; q < 4; ++q) {
        store(lane * 4 + q, run, v[q]);
        run += v[q];
    }
    return __shfl(inc, 63, 64);
}

// a value every lane holds alike, moved to scalar registers
__device__ __forceinline__ uint64_t wl_uniform64(uint64_t v)
{
    // (the builtin returns int: without the casts a low word with bit 31 set would sign-extend into the high one)
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) |
           (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v);
}

// c entries from LDS (s) to global memory (d) by `nl` cooperating lanes (this one is number `sub`; nl >= 8): the part of
// the run that starts on a 16-byte boundary of the destination goes out as 16-byte stores -- a CU's store path moves a
// wave's worth of single dwords at about 11 bytes a clock, and the lists are 16 GB -- the up to three entries before it
// and after it as single stores (one pass: lanes 0..2 the head, lanes 4..6 the tail)
__device__ __forceinline__ void wl_copy_run(uint32_t *d, const uint32_t *s, uint32_t c, uint32_t sub, uint32_t nl)
{
    uint32_t a = (4u - ((uint32_t)((uintptr_t)d >> 2) & 3u)) & 3u;
    if (a > c) a = c;
    const uint32_t nvec = (c - a) >> 2, t0 = a + 4 * nvec;
    if (sub < a) d[sub] = s[sub];
    else if (sub >= 4 && sub - 4 < c - t0) d[t0 + sub - 4] = s[t0 + sub - 4];
    for (uint32_t v = sub; v < nvec; v += nl) {
        const uint32_t p = a + 4 * v;
        wl_v4u x;
        x.x = s[p];
        x.y = s[p + 1];
        x.z = s[p + 2];
        x.w = s[p + 3];
        *reinterpret_cast<wl_v4u *>(d + p) = x;
    }
}

// gbase[g] = first list slot of group g (32 slots per mask word, counted from the batch's first word), g = 0..ngroups
__global__ void wl_gbase_kernel(const uint64_t *__restrict__ mask_off, uint64_t n, uint32_t R, uint32_t ngroups,
                                uint64_t *__restrict__ gbase)
{
    const uint64_t first = mask_off[0];
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g <= ngroups; g += gridDim.x * blockDim.x) {
        const uint64_t r = (uint64_t)g * R < n ? (uint64_t)g * R : n;
        gbase[g] = (mask_off[r] - first) * 32;
    }
}

// ---------------------------------------------------------------------------
// part
// ---------------------------------------------------------------------------
// the reads [r0, r1) of unit u of the chunk that starts with group g_first; tag0 = the first read's index in its group
struct wl_unit {
    uint32_t g, tag0;
    uint64_t r0, r1;
};
__device__ __forceinline__ wl_unit wl_unit_of(uint32_t u, uint32_t P, uint32_t g_first, uint32_t R, uint32_t Ru, uint64_t n)
{
    wl_unit x;
    x.g = g_first + u / P;
    const uint32_t j = u % P;
    const uint64_t r0g = (uint64_t)x.g * R, r1g = r0g + R < n ? r0g + R : n;
    x.r0 = r0g + (uint64_t)j * Ru < r1g ? r0g + (uint64_t)j * Ru : r1g;
    x.r1 = x.r0 + Ru < r1g ? x.r0 + Ru : r1g;
    x.tag0 = (uint32_t)(x.r0 - r0g);
    return x;
}

// walk 1: cnt1[u][s] = windows of unit u in slice s.  A wave per read, a lane per 32-base chunk; LDS counters
// [slice][lane & 31], free of bank conflicts.
__global__ __launch_bounds__(1024) void wl_count_kernel(const uint32_t *__restrict__ codes, const uint32_t *__restrict__ mask,
                                                        const uint64_t *__restrict__ code_off,
                                                        const uint64_t *__restrict__ mask_off,
                                                        const uint32_t *__restrict__ lens, uint64_t n, uint32_t R,
                                                        uint32_t Ru, uint32_t P, uint32_t g_first, uint32_t nunits,
                                                        uint32_t *__restrict__ cnt1)
{
    __shared__ __attribute__((aligned(16))) uint32_t ctr[WL_SLICES * 32];
    const uint32_t tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u;
    for (uint32_t u = blockIdx.x; u < nunits; u += gridDim.x) {
        const wl_unit un = wl_unit_of(u, P, g_first, R, Ru, n);
        lrb_barrier();
        for (uint32_t i = tid; i < WL_SLICES * 32; i += 1024) ctr[i] = 0;
        lrb_barrier();
        const uint32_t col32 = lane & 31u;
        for (uint64_t r = un.r0 + wave; r < un.r1; r += 16) {
            const uint32_t L = lens[r];
            if (L < 15u || L > WL_MAX_WINDOWS + 14u) continue; // over-long reads: the callers' gather / atomic kernels
            const uint32_t *cw = codes + code_off[r];
            const uint32_t *mw = mask + mask_off[r];
            const uint32_t nchunks = (L + 31) >> 5;
            for (uint32_t c = lane; c < nchunks; c += WAVE) {
                const uint32_t vm = valid15_starts(mw[c], mw[c + 1]);
                if (!vm) continue;
                const uint32_t c0 = cw[2 * c], c1 = cw[2 * c + 1], c2 = cw[2 * c + 2];
                const uint32_t q0 = rc32(c0), q1 = rc32(c1), q2 = rc32(c2);
                auto tally1 = [&](int i) {
                    const uint32_t val = i < 16 ? k15_at(c0, c1, i) : k15_at(c1, c2, i - 16);
                    const uint32_t rc = (i < 16 ? __builtin_amdgcn_alignbit(q1, q0, 2 * i)
                                                : __builtin_amdgcn_alignbit(q2, q1, 2 * (i - 16))) & K15_MASK;
                    // the slice is the top 8 bits of the pair index = bits 29..22 of the canonical strand
                    const uint32_t canon = (val & 0x8000u) ? rc : val;
                    atomicAdd(&ctr[((canon >> (WL_SLICE_BITS + 1)) << 5) | col32], 1u);
                };
                if (vm == 0xFFFFFFFFu) {
#pragma unroll
                    for (int i = 0; i < 32; ++i) tally1(i);
                } else {
#pragma unroll
                    for (int i = 0; i < 32; ++i)
                        if (vm & (0x80000000u >> i)) tally1(i);
                }
            }
        }
        // (the barrier that showed why every barrier here is lrb_barrier(): lrb_device.h)
        lrb_barrier();
        {   // four threads per slice, eight lane columns each
            const uint4 *p = reinterpret_cast<const uint4 *>(&ctr[(tid >> 2) * 32 + (tid & 3u) * 8]);
            const uint4 a = p[0], b = p[1];
            uint32_t s = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            if ((tid & 3u) == 0) cnt1[(uint64_t)u * WL_SLICES + (tid >> 2)] = s;
        }
    }
}

// (tests of the count / part check: the first non-zero count of the chunk's first unit made one short)
__global__ void wl_fault_kernel(uint32_t *cnt1)
{
    for (uint32_t i = 0; i < WL_SLICES; ++i)
        if (cnt1[i]) {
            cnt1[i] -= 1;
            return;
        }
}

// Where everything goes.  The (group, slice) lists of the level-1 scratch are CONTIGUOUS -- unit after unit inside a
// slice -- so the order kernel reads one run.  start1[u][s] = where unit u appends its windows of slice s, from the
// group's first slot; bounds[g][64 s] = where the group's slice s starts (the order kernel fills in the buckets),
// bounds[g][16384] = the group's entries.
__global__ __launch_bounds__(256) void wl_gscan_kernel(const uint32_t *__restrict__ cnt1, uint32_t P, uint32_t g_first,
                                                       uint32_t *__restrict__ start1, uint32_t *__restrict__ bounds)
{
    __shared__ uint32_t sz[WL_SLICES], st[WL_SLICES];
    const uint32_t tid = threadIdx.x, gl = blockIdx.x;
    uint32_t c[WL_MAX_UNITS], s = 0;
#pragma unroll
    for (uint32_t j = 0; j < WL_MAX_UNITS; ++j) {
        c[j] = j < P ? cnt1[((uint64_t)gl * P + j) * WL_SLICES + tid] : 0u;
        s += c[j];
    }
    sz[tid] = s;
    lrb_barrier();
    if (tid < 64) {
        uint32_t *bg = bounds + (uint64_t)(g_first + gl) * WL_BSTRIDE;
        const uint32_t total = wl_wave_scan256(tid, [&](uint32_t i) { return sz[i]; }, [&](uint32_t i, uint32_t ex, uint32_t) {
            bg[i * WL_SUBS] = ex;
            st[i] = ex;
        });
        if (tid == 0) bg[WL_BUCKETS] = total;
    }
    lrb_barrier();
    uint32_t run = st[tid];
#pragma unroll
    for (uint32_t j = 0; j < WL_MAX_UNITS; ++j)
        if (j < P) {
            start1[((uint64_t)gl * P + j) * WL_SLICES + tid] = run;
            run += c[j];
        }
}

// ---------------------------------------------------------------------------
// part: the unit's 256 level-1 lists are written through 256 RINGS of 128 entries in LDS, one a slice, whose
// slots are the list positions mod 128 -- shifted so that a ring's blocks of 32 slots are the list's 128-byte lines.  A
// window is placed by ONE returning 64-bit atomic -- {the slice's tail, how far its ring is flushed}: the window's
// position and whether the ring has room for it -- and one store; between two barriers the waves flush every ring's
// complete lines (four lanes a line, 32 bytes each).  No tile is sorted, nothing is scanned, the pair index is made once.
// A slice that draws more than its ring holds between two flushes (>= 97 entries of the 16 k appended; one phase in a
// hundred on uniform reads, most phases on very skewed ones) costs no extra round: a window that finds the ring full
// stores its entry straight into the list (consecutive lanes drew consecutive positions), the flush writes the ring's
// four lines, moves the slice's flush mark to its tail's line and notes where in that line the ring's own entries begin.
// LDS is addressed by byte offsets from 0 (the kernel has no static LDS, its dynamic allocation starts at 0 -- checked
// when the kernel starts): an address is a few bit operations on the window's code, tables are immediate offsets.
// ---------------------------------------------------------------------------
#define WLR_READS 512u    // reads of a unit (wl_units: R / 4 <= 512, R / 2 < 128, R < 64)
#define WLR_TF 0u         // [256] {tail, flushed}: ring positions in BYTES (x 4), eight bytes a slice
#define WLR_LEAD 2048u    // [256] the unit's first position of the slice (bytes)
#define WLR_BASE 3072u    // [256] list byte offset of position 0, from the group's first slot (may wrap)
#define WLR_MOFF 4096u    // [WLR_READS + 1] mask word of a read, from the unit's first
#define WLR_COFF 6176u    // [WLR_READS] u64 code word of a read; bit 63: the read is over-long
#define WLR_RINGS 10272u  // [256][128] u32
#define WLR_SMEM_BYTES (WLR_RINGS + WL_SLICES * 512u)

typedef __attribute__((address_space(3))) uint32_t wl_l32;
typedef __attribute__((address_space(3))) unsigned long long wl_l64;
typedef __attribute__((address_space(3))) wl_v4u wl_l128;
__device__ __forceinline__ uint32_t wl_lds32(uint32_t a) { return *reinterpret_cast<wl_l32 *>(a); }
__device__ __forceinline__ unsigned long long wl_lds64(uint32_t a) { return *reinterpret_cast<wl_l64 *>(a); }
__device__ __forceinline__ wl_v4u wl_lds128(uint32_t a) { return *reinterpret_cast<wl_l128 *>(a); }
__device__ __forceinline__ void wl_lds_set32(uint32_t a, uint32_t v) { *reinterpret_cast<wl_l32 *>(a) = v; }
__device__ __forceinline__ void wl_lds_set64(uint32_t a, unsigned long long v) { *reinterpret_cast<wl_l64 *>(a) = v; }
__device__ __forceinline__ unsigned long long wl_lds_add64(uint32_t a, unsigned long long v)
{
    return __hip_atomic_fetch_add(reinterpret_cast<wl_l64 *>(a), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__global__ __launch_bounds__(1024) void wl_part_kernel(
    const uint32_t *__restrict__ codes, const uint32_t *__restrict__ mask, const uint64_t *__restrict__ code_off,
    const uint64_t *__restrict__ mask_off, const uint32_t *__restrict__ lens, uint64_t n, uint32_t R, uint32_t Ru,
    uint32_t P, uint32_t g_first, uint32_t nunits, const uint64_t *__restrict__ gbase, uint32_t *__restrict__ tmp,
    const uint32_t *__restrict__ start1, const uint32_t *__restrict__ cnt1, uint32_t *__restrict__ mismatch)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t wlr_smem[];
    if ((uint32_t)(uintptr_t)(wl_l32 *)wlr_smem != 0u) __builtin_trap();
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint64_t tmp0 = wl_uniform64(gbase[g_first]);
    for (uint32_t u = blockIdx.x; u < nunits; u += gridDim.x) {
        const wl_unit un = wl_unit_of(u, P, g_first, R, Ru, n);
        const uint64_t r0 = un.r0, r1 = un.r1;
        const uint64_t w0 = wl_uniform64(mask_off[r0]), w1 = wl_uniform64(mask_off[r1]);
        uint32_t *dst = tmp + (wl_uniform64(gbase[un.g]) - tmp0);
        char *dstb = reinterpret_cast<char *>(dst);
        const uint32_t nwords = (uint32_t)(w1 - w0), nreads = (uint32_t)(r1 - r0), rtag0 = un.tag0;
        const uint32_t *umask = mask + w0;
        lrb_barrier();
        if (tid < WL_SLICES) {
            const uint32_t st = start1[(uint64_t)u * WL_SLICES + tid];
            const uint32_t p0 = (((uint32_t)((uintptr_t)dst >> 2) & 31u) + st) & 127u;
            wl_lds_set64(WLR_TF + 8 * tid, (unsigned long long)(p0 * 4u) | ((unsigned long long)((p0 & ~31u) * 4u) << 32));
            wl_lds_set32(WLR_LEAD + 4 * tid, p0 * 4u);
            wl_lds_set32(WLR_BASE + 4 * tid, (st - p0) * 4u);
        }
        if (tid <= nreads && tid <= WLR_READS) {
            const uint64_t r = r0 + tid;
            wl_lds_set32(WLR_MOFF + 4 * tid, tid < nreads ? (uint32_t)(mask_off[r] - w0) : 0xFFFFFFFFu);
            if (tid < nreads) wl_lds_set64(WLR_COFF + 8 * tid, code_off[r] | (lens[r] > WL_MAX_WINDOWS + 14u ? 1ull << 63 : 0ull));
        }
        lrb_barrier();
        // a thread's mask word of a tile of 1,024: its thirty-two window starts, the three code words, the read's tag
        uint32_t cur = 0; // (uniform) the read that holds the first word this wave looked at last
        // A tile's inputs come in two steps, each asked for a whole tile ahead of its use: the mask word pair of tile t + 2
        // and -- through the read table and tile t + 1's mask words -- the code words of tile t + 1 while tile t is
        // appended; both are waited for ONCE, in front of tile t's flush, when they have long arrived (a load waited for
        // behind the flush's stores would wait for those to be acknowledged).
        auto ask_mask = [&](uint32_t wbase, uint32_t &m0, uint32_t &m1) {
            const uint32_t w = wbase + tid;
            m0 = m1 = 0;
            if (w < nwords) m0 = umask[w];
            if (w + 1 < nwords) m1 = umask[w + 1];
        };
        auto ask_codes = [&](uint32_t wbase, uint32_t m0, uint32_t m1, uint32_t &vm, uint32_t &a, uint32_t &b, uint32_t &c, uint32_t &tag) {
            const uint32_t wfirst = wbase + (tid & ~63u), w = wbase + tid;
            vm = tag = 0;
            // (the three words are loaded by EVERY lane, from the batch's first words when the lane has no window: a load
            // inside the branch is followed by copies into the registers the other path zeroes, i.e. by a wait for it)
            const uint32_t *cw = codes;
            if (wfirst < nwords) {
                for (;;) { // the reads that start at or before the wave's first word, sixty-four at a time
                    const uint32_t j = cur + 1 + lane;
                    const uint32_t k = (uint32_t)__popcll(__ballot(j < nreads && wl_lds32(WLR_MOFF + 4 * j) <= wfirst));
                    cur += k;
                    if (k < 64) break;
                }
                cur = __builtin_amdgcn_readfirstlane(cur);
                const uint32_t v = m0 ? valid15_starts(m0, m1) : 0u;
                if (v) {
                    uint32_t jl = cur;
                    while (wl_lds32(WLR_MOFF + 4 * (jl + 1)) <= w) ++jl; // (the entry behind the last read is above every word)
                    const uint64_t cj = wl_lds64(WLR_COFF + 8 * jl);
                    if (!(cj >> 63)) {
                        vm = v;
                        cw = codes + cj + 2 * (w - wl_lds32(WLR_MOFF + 4 * jl));
                        tag = (rtag0 + jl) << WL_SLICE_BITS;
                    }
                }
            }
            a = cw[0];
            b = cw[1];
            c = cw[2];
        };
        // the canonical strand of a window: its top eight bits are the slice, the pair index drops bit 15
        auto canon = [&](uint32_t val, uint32_t rc) { return (val & 0x8000u) ? rc : val; };
        auto entry_of = [&](uint32_t x, uint32_t tag) { return ((((x >> 1) & ~0x7FFFu) | (x & 0x7FFFu)) & WL_OFF_MASK) | tag; };
        auto tf_of = [&](uint32_t x) { return (x >> 19) & 0x7F8u; };                                   // WLR_TF + 8 slice
        auto slot_of = [&](uint32_t x, uint32_t p4) { return (((x >> 13) & 0x1FE00u) | (p4 & 0x1FCu)) + WLR_RINGS; }; // 512 slice + position
        // the ready lines of this wave's sixteen slices: four lanes a line, 32 bytes each
        auto flush_lines = [&]() {
            const uint32_t s = wave * 16 + (lane >> 2), q32 = (lane & 3u) * 32u;
            const unsigned long long tfv = wl_lds64(WLR_TF + 8 * s);
            const uint32_t f4 = (uint32_t)(tfv >> 32), te4 = (uint32_t)tfv, t4 = te4 & ~127u;
            const uint32_t ld4 = wl_lds32(WLR_LEAD + 4 * s), b4 = wl_lds32(WLR_BASE + 4 * s);
            // (a tail beyond the ring's four lines: what lies there was stored straight into the list by the windows themselves)
            const bool past = te4 - f4 > 512u;
            const uint32_t lim4 = past ? f4 + 512u : t4;
            const uint32_t nb = (lim4 - f4) >> 7;
            for (uint32_t k = 0; k < 4; ++k) {
                if (!__ballot(k < nb)) break;
                if (k < nb) {
                    const uint32_t bp4 = f4 + 128 * k + q32;
                    const uint32_t la = WLR_RINGS + s * 512u + (bp4 & 0x1FFu);
                    const wl_v4u v0 = wl_lds128(la), v1 = wl_lds128(la + 16);
                    const uint32_t off = b4 + bp4;
                    if (bp4 < ld4) { // the line holds entries that are not this ring's: its own go out one by one
                        const uint32_t e[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                        for (uint32_t i = 0; i < 8; ++i)
                            if (bp4 + 4 * i >= ld4) reinterpret_cast<uint32_t *>(dstb + off)[i] = e[i];
                    } else {
                        reinterpret_cast<wl_v4u *>(dstb + off)[0] = v0;
                        reinterpret_cast<wl_v4u *>(dstb + off)[1] = v1;
                    }
                }
            }
            wl_wave_lds_fence();
            if ((lane & 3u) == 0) {
                if (nb) wl_lds_set32(WLR_TF + 8 * s + 4, t4);
                if (past) wl_lds_set32(WLR_LEAD + 4 * s, te4); // the tail's open line starts with entries that are in the list already
            }
        };
        uint32_t vm, a, b, c, tag, m0n, m1n;
        {
            uint32_t m0, m1;
            ask_mask(0, m0, m1);
            ask_codes(0, m0, m1, vm, a, b, c, tag);
            ask_mask(1024, m0n, m1n);
            asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(m0n), "+v"(m1n));
        }
        for (uint32_t wbase = 0; wbase < nwords; wbase += 1024) {
            uint32_t vmn, an, bn, cn, tagn, m0nn, m1nn;
            ask_codes(wbase + 1024, m0n, m1n, vmn, an, bn, cn, tagn);
            ask_mask(wbase + 2048, m0nn, m1nn);
            const uint32_t ra = rc32(a), rb = rc32(b), rc = rc32(c);
            auto window = [&](int i) {
                return i < 16 ? canon(k15_at(a, b, i), __builtin_amdgcn_alignbit(rb, ra, 2 * i) & K15_MASK)
                              : canon(k15_at(b, c, i - 16), __builtin_amdgcn_alignbit(rc, rb, 2 * (i - 16)) & K15_MASK);
            };
            // a window with its position: into the ring when the ring has room (the flush mark came back with the position),
            // else -- rarely on uniform reads, mostly for the slices of a homopolymer -- straight to its place in the list:
            // consecutive lanes drew consecutive positions, the flush leaves those lines alone
            auto place = [&](uint32_t x, uint32_t p4, uint32_t f4) {
                if (p4 - f4 < 512u) wl_lds_set32(slot_of(x, p4), entry_of(x, tag));
                else *reinterpret_cast<uint32_t *>(dstb + (uint32_t)(wl_lds32(WLR_BASE + (tf_of(x) >> 1)) + p4)) = entry_of(x, tag);
            };
#pragma unroll
            for (int h = 0; h < 32; h += 16) {
                // sixteen windows a thread, then a flush.  One path for the whole wave: the walk with every window counting
                // when all its lanes have sixteen here, else the same walk with a lane mask a window -- never the two one
                // after the other, and never a chain of dependent LDS round trips: the slowest wave is what the barrier
                // waits for.
                const bool all_full = __ballot(((vm << h) >> 16) != 0xFFFFu) == 0;
                if (all_full) {
                    // four windows at a time, the next four made while the LDS works on these
                    uint32_t x[16];
                    unsigned long long r[16];
                    auto issue = [&](int j0) {
#pragma unroll
                        for (int j = j0; j < j0 + 4; ++j) r[j] = wl_lds_add64(tf_of(x[j]), 4ull);
                    };
                    auto finish = [&](int j0) {
#pragma unroll
                        for (int j = j0; j < j0 + 4; ++j) place(x[j], (uint32_t)r[j], (uint32_t)(r[j] >> 32));
                    };
#pragma unroll
                    for (int j = 0; j < 4; ++j) x[j] = window(h + j);
                    issue(0);
#pragma unroll
                    for (int q = 4; q < 16; q += 4) {
#pragma unroll
                        for (int j = q; j < q + 4; ++j) x[j] = window(h + j);
                        __builtin_amdgcn_sched_barrier(0);
                        issue(q);
                        __builtin_amdgcn_sched_barrier(0);
                        finish(q - 4);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    finish(12);
                } else {
#pragma unroll
                    for (int q = 0; q < 16; q += 8) { // (eight at a time: sixteen positions in flight would not fit the registers)
                        uint32_t x[8];
                        unsigned long long r[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) x[j] = window(h + q + j);
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            if (vm & (0x80000000u >> (h + q + j))) r[j] = wl_lds_add64(tf_of(x[j]), 4ull);
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            if (vm & (0x80000000u >> (h + q + j))) place(x[j], (uint32_t)r[j], (uint32_t)(r[j] >> 32));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                lrb_barrier(); // every window of the phase is in its ring or in the list
                if (h == 0) // (the next tile's words, asked for at the top of this one: waited for once, in front of the flush's stores)
                    asm volatile("" : "+v"(an), "+v"(bn), "+v"(cn), "+v"(m0nn), "+v"(m1nn));
                flush_lines();
                lrb_barrier(); // the rings are empty but for their tails' open lines
            }
            vm = vmn;
            a = an;
            b = bn;
            c = cn;
            tag = tagn;
            m0n = m0nn;
            m1n = m1nn;
        }
        // what is left in the rings: the unit's last, partial lines (thirty-two lanes a slice)
#pragma unroll
        for (uint32_t p = 0; p < 8; ++p) {
            const uint32_t s = wave * 16 + p * 2 + (lane >> 5);
            const unsigned long long tfv = wl_lds64(WLR_TF + 8 * s);
            const uint32_t pp4 = (uint32_t)(tfv >> 32) + 4 * (lane & 31u);
            if (pp4 < (uint32_t)tfv && pp4 >= wl_lds32(WLR_LEAD + 4 * s))
                *reinterpret_cast<uint32_t *>(dstb + (uint32_t)(wl_lds32(WLR_BASE + 4 * s) + pp4)) = wl_lds32(WLR_RINGS + s * 512u + (pp4 & 0x1FFu));
        }
        // The unit appended to every slice EXACTLY what the count kernel counted for it?  (round 6: with several processes
        // time-sliced on one GPU the count came out one short for one (unit, slice) in about one partition of a hundred
        // -- the part kernel then runs one entry into the next run and a window is lost, silently.  The two walks are
        // the same arithmetic on the same words, so a disagreement is not the data's: the host repeats the partition.)
        if (tid < WL_SLICES) {
            const uint32_t st = start1[(uint64_t)u * WL_SLICES + tid];
            const uint32_t p0 = (((uint32_t)((uintptr_t)dst >> 2) & 31u) + st) & 127u;
            const uint32_t te4 = (uint32_t)wl_lds64(WLR_TF + 8 * tid);
            if (((te4 - p0 * 4u) >> 2) != cnt1[(uint64_t)u * WL_SLICES + tid]) atomicOr(mismatch, 1u);
        }
    }
}

// ---------------------------------------------------------------------------
// order: grid (256 slices, groups of the chunk)
// ---------------------------------------------------------------------------
#define WL_ORDER_CACHE 64 // entries a thread keeps in registers: lists of up to 65,536 entries are read once
// A workgroup walks the lists of one slice for groups blockIdx.y, blockIdx.y + gridDim.y, ...  With CACHED the NEXT
// list is asked for while the current one is ordered: the sixteen registers a thread holds of a tile are free once the
// tile's entries have their places, and the loads of the next list's rows go into them -- unconditional buffer loads
// through a resource cut to the list (no next list, or one too long for the registers: zero records, no memory access),
// so that the list's way from HBM lies under the copy-outs instead of in front of the tally.
template <bool CACHED>
__device__ __forceinline__ void wl_order_body(const uint32_t *__restrict__ tmp, uint32_t g_first, uint32_t ngroups,
                                              uint32_t min_total, const uint64_t *__restrict__ gbase,
                                              uint32_t *__restrict__ lists, uint32_t *__restrict__ bounds,
                                              const uint32_t *__restrict__ bounds_ro)
{
    __shared__ __attribute__((aligned(16))) uint32_t sorted[WL_TILE];
    __shared__ uint32_t tot[WL_SUBS * 16]; // pass A: [bucket][lane column 16] u32
    __shared__ uint32_t ctr[1024];         // tiles: [bucket 64][lane column 16]
    __shared__ uint32_t ctr4[CACHED ? 4096 : 1]; // the same for the four tiles of a list held in registers
    __shared__ uint32_t cnt[WL_SUBS], lbase[WL_SUBS], gcur[WL_SUBS];
    const uint32_t tid0 = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const uint32_t sl = blockIdx.x;
    const uint64_t goff0 = gbase[g_first];
    constexpr uint32_t CAP = CACHED ? WL_ORDER_CACHE * 1024u : 0u;
    // (slot below: one u32 counter per (bucket, lane & 15), no half-word arithmetic around the atomics)
    // the group's level-1 list of this slice: one run of the scratch buffer, from where the group's slice starts
    // (bounds[g][64 sl], which the scan kernel wrote and this kernel leaves as it is) to where the next one does
    struct list_t {
        uint32_t gstart, total;
        uint64_t goff;
    };
    auto list_of = [&](uint32_t g, bool there) {
        list_t l = {0, 0, goff0};
        if (there) {
            // (the slices' first words are the scan kernel's and nobody writes them here: read through bounds_ro, the
            // same array as a read-only kernel argument, i.e. as scalar loads -- a vector load's wait would also be a
            // wait for every row asked for before it)
            const uint32_t *bg = bounds_ro + (uint64_t)g * WL_BSTRIDE + sl * WL_SUBS;
            l.gstart = bg[0];
            l.total = bg[WL_SUBS] - l.gstart;
            l.goff = gbase[g];
        }
        return l;
    };
    auto source = [&](const list_t &l) { return tmp + (l.goff - goff0) + l.gstart; };
    // rows q0 .. q0 + nq - 1 (1,024 entries each) of a list into the registers; nothing is read past the list's end
    // The FULL rows only; the last, partial row has a register of its own (elast) -- with it among the rows every row
    // would carry a lane mask from the tally to the ranks, sixty-four of them, in SGPRs the kernel does not have.
    uint32_t e[CACHED ? WL_ORDER_CACHE : 1], elast = 0;
    auto ask = [&](const list_t &l, int q0, int nq) {
        if (!CACHED) return;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint32_t *>(source(l)), 0, (int)(l.total <= CAP ? (l.total & ~1023u) * 4u : 0u), 0x00020000);
        // (the byte offset goes through the VGPR operand, the only one the range check covers; laundered so that the
        // sixty-four sums are made where they are used instead of being kept in registers across the loop)
        uint32_t t4 = tid0 * 4u;
        asm volatile("" : "+v"(t4));
#pragma unroll
        for (int q = q0; q < q0 + nq; ++q) e[q] = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(t4 + q * 4096u), 0, 0);
    };
    auto ask_last = [&](const list_t &l) {
        if (!CACHED) return;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint32_t *>(source(l)), 0, (int)(l.total <= CAP ? l.total * 4u : 0u), 0x00020000);
        elast = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)((l.total & ~1023u) * 4u + tid0 * 4u), 0, 0);
    };
    const uint32_t gend = g_first + ngroups;
    uint32_t g = g_first + blockIdx.y;
    list_t cur = list_of(g, g < gend);
    ask(cur, 0, WL_ORDER_CACHE);
    ask_last(cur);
    for (; g < gend; g += gridDim.y) {
        const list_t nxt = list_of(g + gridDim.y, g + gridDim.y < gend);
        // (the thread index laundered per list: what is derived from it is made again for every list -- a few VALU
        // instructions -- instead of being kept across the loop in registers the list itself needs)
        uint32_t tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const uint32_t lane = tid & 63u, c16 = tid & 15u;
        auto slot = [&](uint32_t ev) { return (((ev >> WL_SUB_BITS) & 63u) << 4) | c16; };
        const uint32_t gstart = cur.gstart, total = cur.total;
        uint32_t *bg = bounds + (uint64_t)g * WL_BSTRIDE + sl * WL_SUBS;
        const uint32_t *src = source(cur);
        uint32_t *dst = lists + cur.goff + gstart;
        if (!CACHED && total <= min_total && min_total) {
            // (the companion launch of the register kernel: that one has ordered this list)
        } else if (total == 0 || (CACHED && total > CAP)) {
            // nothing to order here: an empty list, or one too long for the registers (the streamed kernel's, launched
            // behind this one -- with its code in this loop the compiler spills the list's registers)
            if (total == 0 && tid > 0 && tid < WL_SUBS) bg[tid] = gstart;
            ask(nxt, 0, WL_ORDER_CACHE);
            ask_last(nxt);
        } else if constexpr (CACHED) {
            // The whole list in registers (WL_ORDER_CACHE entries a thread): read ONCE, and tallied ONCE -- per tile of
            // 16 k entries, bucket and lane column -- so that one scan gives every tile's places in LDS and in the list;
            // then a tile costs its rank atomics, the scattered LDS stores and the copy.
            constexpr int NTL = WL_ORDER_CACHE / 16;
            static_assert(NTL == 4, "the scan below walks four tiles");
            uint32_t *cnt4 = tot; // [tile][bucket] counts, then (from word 256) [tile][bucket] {LDS start, list start} pairs
            // rows of 1,024 entries: a full row needs no test per lane (a uniform branch); the last row is elast
            const uint32_t nfull = total >> 10, tlast = nfull >> 4;
            const bool in_last = tid < (total & 1023u);
#pragma unroll
            for (int t = 0; t < NTL; ++t) ctr4[t * 1024 + tid] = 0;
            // every row has to be here now (vmcnt 0, said aloud: left to the compiler the wait sits behind the rows'
            // length tests, and a list without full rows would leave the registers "in flight" for the tiles below,
            // whose waits would then be waits for the NEXT list's rows)
            __builtin_amdgcn_s_waitcnt(0x0F70);
            lrb_barrier();
#pragma unroll
            for (int q = 0; q < WL_ORDER_CACHE; ++q)
                if (q < nfull) atomicAdd(&ctr4[(q / 16) * 1024 + slot(e[q])], 1u);
            if (in_last) atomicAdd(&ctr4[tlast * 1024 + slot(elast)], 1u);
            lrb_barrier();
            // thread tid owns word tid of every tile (bucket tid >> 4, lane column tid & 15); the counts of two tiles ride
            // one register through the prefix over the sixteen columns (a tile's bucket holds at most 16,384 entries)
            uint32_t kk[NTL], ex[NTL];
#pragma unroll
            for (int t = 0; t < NTL; ++t) kk[t] = ctr4[t * 1024 + tid];
#pragma unroll
            for (int h = 0; h < NTL / 2; ++h) {
                const uint32_t k2 = kk[2 * h] | (kk[2 * h + 1] << 16);
                const uint32_t inc = wl_group_scan_incl<16>(k2, lane);
                if ((tid & 15u) == 15u) {
                    cnt4[(2 * h) * 64 + (tid >> 4)] = inc & 0xFFFFu;
                    cnt4[(2 * h + 1) * 64 + (tid >> 4)] = inc >> 16;
                }
                const uint32_t e2 = inc - k2;
                ex[2 * h] = e2 & 0xFFFFu;
                ex[2 * h + 1] = e2 >> 16;
            }
            lrb_barrier();
            {   // every wave: lane l = bucket l.  Per tile the exclusive scan over buckets (LDS starts); over the tiles'
                // sums the list starts
                uint32_t c[NTL], lb[NTL], sum = 0;
#pragma unroll
                for (int t = 0; t < NTL; ++t) {
                    c[t] = cnt4[t * 64 + lane];
                    sum += c[t];
                    lb[t] = wl_wave_scan_incl(c[t]) - c[t];
                }
                const uint32_t base = wl_wave_scan_incl(sum) - sum;
#pragma unroll
                for (int t = 0; t < NTL; ++t) // (a thread's own words: nobody else reads them before the barrier below)
                    ctr4[t * 1024 + tid] = __shfl(lb[t], wave * 4 + (lane >> 4), 64) + ex[t];
                if (wave == 0) {
                    // (word 0 is the scan kernel's and is read through bounds_ro by this workgroup and its neighbour: the
                    // value is the same, the store is not made)
                    if (lane > 0) bg[lane] = gstart + base;
                    uint32_t run = base;
#pragma unroll
                    for (int t = 0; t < NTL; ++t) {
                        cnt4[256 + (t * 64 + lane) * 2] = lb[t];
                        cnt4[256 + (t * 64 + lane) * 2 + 1] = run;
                        run += c[t];
                    }
                }
            }
            lrb_barrier();
            // the tiles in turn: ranks (the atomics return them), scattered stores into the tile's sorted image, the next
            // list's rows asked for into the registers this tile has just given up, runs out
            uint32_t ccar[4] = {0u, 0u, 0u, 0u}, crar[4] = {0u, 0u, 0u, 0u}; // per bucket of the wave: entries waiting, how many
#pragma unroll
            for (int t = 0; t < NTL; ++t) {
                const bool live = t * WL_TILE < total;
                if (live) {
                    // (the row count through an SGPR the compiler cannot match with the tally's: it would keep that
                    // pass's sixty-four comparisons as lane masks otherwise, spilled)
                    uint32_t nf = nfull;
                    asm volatile("" : "+s"(nf));
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        if (t * 16 + q < nf) sorted[atomicAdd(&ctr4[t * 1024 + slot(e[t * 16 + q])], 1u)] = e[t * 16 + q];
                    if (t == tlast && in_last) sorted[atomicAdd(&ctr4[t * 1024 + slot(elast)], 1u)] = elast;
                }
                ask(nxt, t * 16, 16);
                if (t == NTL - 1) ask_last(nxt);
                if (live) {
                    lrb_barrier();
                    {   // a wave appends the runs of its four buckets
                        const uint32_t b0 = wave * 4;
                        uint32_t cv = 0, lv = 0, gv = 0;
                        if (lane < 4) {
                            cv = cnt4[t * 64 + b0 + lane];
                            lv = cnt4[256 + (t * 64 + b0 + lane) * 2];
                            gv = cnt4[256 + (t * 64 + b0 + lane) * 2 + 1];
                        }
                        // A bucket's runs of the list's four tiles follow each other in the list, and each ends anywhere: a
                        // 128-byte line written in two pieces a tile apart leaves the L2 twice, as partial writes, at 3.3 TB/s
                        // where whole lines go at 5.4 (scripts/ubench_scatter_write.hip).  So what lies past a bucket's last
                        // line boundary (up to 31 entries) waits in a register, a lane each, and goes out in front of the next
                        // tile's run; the list's last tile writes everything.  (All of this is wave-uniform: SGPRs.)
                        const bool last_tile = (uint32_t)(t + 1) * WL_TILE >= total;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const uint32_t c = (uint32_t)__builtin_amdgcn_readlane(cv, i), l = (uint32_t)__builtin_amdgcn_readlane(lv, i);
                            const uint32_t g = (uint32_t)__builtin_amdgcn_readlane(gv, i);
                            const uint32_t cc = ccar[i], tot_b = cc + c, gp = g - cc; // the carry sits in front of the run
                            const uint32_t to_line = (32u - (((uint32_t)((uintptr_t)dst >> 2) + gp) & 31u)) & 31u;
                            const uint32_t W = last_tile ? tot_b : tot_b >= to_line ? to_line + ((tot_b - to_line) & ~31u) : 0u;
                            if (W) { // (W >= cc: the whole carry goes out)
                                if (lane < cc) dst[gp + lane] = crar[i];
                                wl_copy_run(dst + g, sorted + l, W - cc, lane, 64);
                                if (lane < tot_b - W) crar[i] = sorted[l + (W - cc) + lane];
                                ccar[i] = tot_b - W;
                            } else { // not a line yet: the run joins the carry
                                if (lane >= cc && lane < tot_b) crar[i] = sorted[l + lane - cc];
                                ccar[i] = tot_b;
                            }
                        }
                    }
                    lrb_barrier();
                }
            }
        } else {
            // ---- a longer list (repeats, low-complexity reads): streamed twice.  Pass A: bucket sizes
            tot[tid] = 0;
            ctr[tid] = 0;
            lrb_barrier();
            for (uint32_t i0 = 0; i0 < total; i0 += 8192) {
                uint32_t f[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const uint32_t i = i0 + q * 1024 + tid;
                    f[q] = i < total ? src[i] : 0xFFFFFFFFu;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (i0 + q * 1024 + tid < total) atomicAdd(&tot[slot(f[q])], 1u);
            }
            lrb_barrier();
            if (tid < 64) { // bucket sizes of the whole list -> where each bucket starts (gcur, bounds)
                uint32_t sz = 0;
#pragma unroll
                for (uint32_t q = 0; q < 16; ++q) sz += tot[tid * 16 + ((q + tid) & 15u)];
                const uint32_t inc = wl_wave_scan_incl(sz);
                gcur[tid] = inc - sz;
                if (tid > 0) bg[tid] = gstart + inc - sz; // (word 0: the scan kernel's, read through bounds_ro)
            }
            // ---- pass B: 16 k-entry tiles sorted by bucket in LDS (lane-private counters), runs appended
            uint32_t stale = 0;
            for (uint32_t t0 = 0; t0 < total; t0 += WL_TILE) {
                uint32_t f[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const uint32_t i = t0 + q * 1024 + tid;
                    f[q] = i < total ? src[i] : 0u;
                }
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    if (t0 + q * 1024 + tid < total) atomicAdd(&ctr[slot(f[q])], 1u);
                lrb_barrier(); // B
                // sixteen threads per bucket, one lane column each
                const uint32_t k = ctr[tid] - stale;
                const uint32_t inc = wl_group_scan_incl<16>(k, lane);
                if ((tid & 15u) == 15u) cnt[tid >> 4] = inc;
                const uint32_t ex = inc - k;
                lrb_barrier(); // C
                {
                    const uint32_t v = cnt[lane];
                    const uint32_t lb = wl_wave_scan_incl(v) - v;
                    if (wave == 0) lbase[lane] = lb;
                    const uint32_t st = __shfl(lb, wave * 4 + (lane >> 4), 64) + ex; // this thread's bucket = tid >> 4
                    ctr[tid] = st;
                    stale = st + k;
                }
                lrb_barrier(); // D
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    if (t0 + q * 1024 + tid < total) sorted[atomicAdd(&ctr[slot(f[q])], 1u)] = f[q];
                lrb_barrier(); // E
                {   // a wave appends the runs of its four buckets
                    const uint32_t b0 = wave * 4;
                    uint32_t cv = 0, lv = 0, gv = 0;
                    if (lane < 4) {
                        cv = cnt[b0 + lane];
                        lv = lbase[b0 + lane];
                        gv = gcur[b0 + lane];
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        wl_copy_run(dst + (uint32_t)__builtin_amdgcn_readlane(gv, i),
                                    sorted + (uint32_t)__builtin_amdgcn_readlane(lv, i),
                                    (uint32_t)__builtin_amdgcn_readlane(cv, i), lane, 64);
                    if (lane < 4) gcur[b0 + lane] = gv + cv;
                }
                // (the next tile's tallies touch ctr only; sorted, cnt and lbase are rewritten behind its barriers)
            }
            lrb_barrier();
        }
        cur = nxt;
    }
}

// One workgroup per CU with the list in registers (wl_order_kernel_occ1), and behind it this one for the lists too long for
// that: two workgroups per CU, the list streamed twice (64 VGPRs)
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) void wl_order_kernel(
    const uint32_t *__restrict__ tmp, uint32_t g_first, uint32_t ngroups, uint32_t min_total,
    const uint64_t *__restrict__ gbase, uint32_t *__restrict__ lists, uint32_t *__restrict__ bounds,
    const uint32_t *__restrict__ bounds_ro)
{
    wl_order_body<false>(tmp, g_first, ngroups, min_total, gbase, lists, bounds, bounds_ro);
}

__global__ __launch_bounds__(1024) void wl_order_kernel_occ1(const uint32_t *__restrict__ tmp, uint32_t g_first,
                                                             uint32_t ngroups, const uint64_t *__restrict__ gbase,
                                                             uint32_t *__restrict__ lists, uint32_t *__restrict__ bounds,
                                                             const uint32_t *__restrict__ bounds_ro)
{
    wl_order_body<true>(tmp, g_first, ngroups, 0, gbase, lists, bounds, bounds_ro);
}

// ---------------------------------------------------------------------------
// tally (K2): a workgroup per bucket, the bucket's 2^15 counters in LDS
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void wl_tally_kernel(const uint32_t *__restrict__ lists,
                                                        const uint32_t *__restrict__ bounds,
                                                        const uint64_t *__restrict__ gbase, uint32_t ngroups,
                                                        uint32_t *__restrict__ half)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[]; // 32768 counters
    const uint32_t tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u;
    const uint32_t b = blockIdx.x;
    for (uint32_t i = tid; i < 8192u; i += 1024) reinterpret_cast<uint4 *>(hist)[i] = make_uint4(0, 0, 0, 0);
    lrb_barrier();
    uint32_t any = 0;
    // A piece = up to 1,024 entries of one group's run of this bucket: sixteen loads a lane, through a buffer
    // resource cut to the piece (lanes past its end read zeros without a memory access, so every load is
    // unconditional and the compiler can count them: the next piece's sixteen stay in flight while this one's are
    // tallied)
    auto ask = [&](const uint32_t *src, uint32_t len, uint32_t *e) {
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(src), 0, (int)(len * 4u), 0x00020000);
#pragma unroll
        for (int q = 0; q < 16; ++q) e[q] = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(lane * 4u), q * 256, 0);
    };
    auto tally = [&](uint32_t len, const uint32_t *e) {
#pragma unroll
        for (int q = 0; q < 16; ++q)
            if (q * 64 + lane < len) atomicAdd(&hist[e[q] & 0x7FFFu], 1u);
    };
    // wave w walks groups w, w + 16, ...; lane j fetches the bounds of the j-th of them (64 groups per pass)
    for (uint32_t gq = 0; gq < ngroups; gq += 1024) {
        const uint32_t g = gq + lane * 16 + wave;
        uint32_t b0 = 0, b1 = 0;
        uint64_t base = 0;
        if (g < ngroups) {
            b0 = bounds[(uint64_t)g * WL_BSTRIDE + b];
            b1 = bounds[(uint64_t)g * WL_BSTRIDE + b + 1];
            base = gbase[g];
        }
        const uint32_t rem = ngroups - gq; // groups of this wave in the pass
        const uint32_t nj = rem > wave ? ((rem - wave + 15) / 16 < 64 ? (rem - wave + 15) / 16 : 64) : 0;
        // the pieces of the wave's groups in turn (uniform state: group j, offset within its run)
        uint32_t j = 0, off = 0;
        auto piece = [&](const uint32_t *&src, uint32_t &len) { // the current piece, then step to the next
            src = lists;
            len = 0;
            while (j < nj) {
                const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane(b0, j), s1 = (uint32_t)__builtin_amdgcn_readlane(b1, j);
                if (s0 + off < s1) {
                    const uint64_t sb = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((uint32_t)(base >> 32), j) << 32) |
                                        (uint32_t)__builtin_amdgcn_readlane((uint32_t)base, j);
                    src = lists + sb + s0 + off;
                    len = s1 - s0 - off < 1024u ? s1 - s0 - off : 1024u;
                    off += 1024;
                    return;
                }
                ++j;
                off = 0;
            }
        };
        uint32_t ea[16], eb[16];
        const uint32_t *sa, *sb_;
        uint32_t la, lb;
        piece(sa, la);
        ask(sa, la, ea);
        while (la) {
            piece(sb_, lb);
            ask(sb_, lb, eb);
            any |= la;
            tally(la, ea);
            if (!lb) break;
            piece(sa, la);
            ask(sa, la, ea);
            tally(lb, eb);
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);      // (lgkmcnt(0): the tallies above are in the counters -- lrb_barrier(), lrb_device.h)
    if (!__syncthreads_or(any != 0)) return; // an untouched bucket costs nothing
    uint4 *t = reinterpret_cast<uint4 *>(half + ((uint64_t)b << WL_SUB_BITS));
    for (uint32_t i = tid; i < 8192u; i += 1024) {
        const uint4 v = reinterpret_cast<const uint4 *>(hist)[i];
        if (v.x | v.y | v.z | v.w) {
            uint4 o = t[i];
            o.x += v.x;
            o.y += v.y;
            o.z += v.z;
            o.w += v.w;
            t[i] = o;
        }
    }
}

// ---------------------------------------------------------------------------
// sweep (K3): a workgroup per group; LDS = [map bucket 32 KB][histograms]
// ---------------------------------------------------------------------------
// A step = one bucket: its piece of the map is in LDS, the group's entries of the bucket are tallied.  A step is short
// (some 900 entries at 1,563 reads a group) and there are 16,384 of them.  Four LOADER waves stage the map (a bucket =
// NP pieces of 4 KB, one 16-byte load a lane each; asked for MD - 1 steps ahead into register sets) and EW ENTRY waves
// walk the lists (up to WL_RING_ENTRIES / (64 EW) entries a lane and step, asked for eight steps ahead).  The roles are
// separate waves because a wave's loads retire IN ORDER (s_waitcnt vmcnt counts them): a wave that waited for the bucket
// it asked for two steps ago would also wait for the entries it asked for a moment ago.  How many entry waves is what the
// kernel's time hangs on (their instruction streams are the critical path of a step: 4 waves 11.5 ms per 4e9 windows, 8:
// 8.5, 12: 8.2); the first form, sixteen waves doing both jobs, took 17.1.  Every load of the step loops is
// unconditional -- list entries through a buffer resource cut to the bucket's end (lanes past it read zeros without a
// memory access), clamped indices elsewhere -- so that the compiler can count the loads in flight and wait for exactly
// the ones it needs; behind a branch it would wait for all of them.
#define WL_SWEEP_DEPTH 8      // steps between the load of a bucket's list entries and their use
#define WL_RING_ENTRIES 1536u // entries of a bucket the ring covers; a longer bucket's rest is read where it is used
// The map as the sweep reads it when the histogram has at most 32 bins: BITS (5 or 4) bits a pair instead of a byte,
// 32 / BITS pairs to a word, a bucket of 2^15 pairs in NP pieces of 4 KB (six / four instead of eight) -- every CU moves the
// whole map through its L2 -> CU path once per round, and that path is what the sweep waits for.  Packed from the byte map
// before every sweep (0.9 GB of traffic: 0.2 ms).
template <int BITS, int NP>
__global__ __launch_bounds__(256) void wl_map_pack_kernel(const uint8_t *__restrict__ map, uint32_t *__restrict__ packed)
{
    constexpr uint32_t PER = 32 / BITS, WPB = NP * 1024;
    const uint64_t total = (uint64_t)WL_BUCKETS * WPB;
    for (uint64_t gw = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; gw < total; gw += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t b = (uint32_t)(gw / WPB), j = (uint32_t)(gw % WPB);
        uint32_t w = 0;
        if (j * PER < 32768u) {
            const uint8_t *src = map + ((uint64_t)b << WL_SUB_BITS) + j * PER;
#pragma unroll
            for (uint32_t t = 0; t < PER; ++t)
                if (j * PER + t < 32768u) w |= (uint32_t)(src[t] & ((1u << BITS) - 1u)) << (BITS * t);
        }
        packed[gw] = w;
    }
}

// DB: TWO buckets of the map in LDS (when the histograms leave room): the loaders write bucket st + 1 while the entry
// waves tally bucket st, one barrier a step instead of two
template <int LW, int EW, int MD, bool DB, int BITS, int NP> // loader waves, entry waves, register sets of a loader wave, map form
__global__ __launch_bounds__(64 * (LW + EW)) void wl_sweep_kernel(const uint32_t *__restrict__ lists,
                                                                 const uint32_t *__restrict__ bounds,
                                                                 const uint64_t *__restrict__ gbase, uint64_t n,
                                                                 uint32_t R, uint32_t ngroups,
                                                                 const uint8_t *__restrict__ map, uint32_t bins,
                                                                 uint32_t *__restrict__ hist_out,
                                                                 uint32_t *__restrict__ sums_out)
{
    constexpr uint32_t NT = 64 * (LW + EW), ET = 64 * EW, NE = (WL_RING_ENTRIES + ET - 1) / ET;
    static_assert(LW == 4, "a piece = one 16-byte load of each lane of the four loader waves = 4 KB");
    constexpr uint32_t BB = NP * 4096, B4 = BB / 16; // bytes / 16-byte vectors of a bucket as staged (8 pieces: the byte map)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
    uint8_t *map_s = smem_raw;
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem_raw + (DB ? 2 * BB : BB));
    const uint32_t tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u;
    // counter (read r, bin b) = u16 half (r & 1) of word b * Rh + (r >> 1): neighbouring reads in neighbouring banks
    const uint32_t Rh = (R + 1) >> 1, hwords = Rh * bins;
    const wl_v4u *map4 = reinterpret_cast<const wl_v4u *>(map);
    for (uint32_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const uint64_t r0 = (uint64_t)g * R, r1 = r0 + R < n ? r0 + R : n;
        const uint32_t *bg = bounds + (uint64_t)g * WL_BSTRIDE;
        const uint32_t *lg = lists + wl_uniform64(gbase[g]);
        lrb_barrier();
        for (uint32_t i = tid; i < hwords; i += NT) hist[i] = 0;
        for (uint32_t i = tid; i < B4; i += NT) reinterpret_cast<wl_v4u *>(map_s)[i] = map4[i];
        lrb_barrier();
        if (wave < LW) {
            // ---- loader: lane lt's NP 16-byte pieces of a bucket are LW KB apart (1 KB per wave-instruction)
            const uint32_t lt = wave * 64 + lane;
            const wl_v4u *mrow = map4 + lt;
            wl_v4u *md = reinterpret_cast<wl_v4u *>(map_s) + lt;
            wl_v4u ms[MD][NP];
            // (the 32 workgroups of an XCD ask their L2 for the same bucket at about the same time; starting each at another
            // piece -- rotated by its number within the XCD -- changed nothing: 11.89 against 11.92 ms)
            // bucket b lives in set b % MD from MD - 1 steps before it is written to LDS
#pragma unroll
            for (int k = 1; k < MD; ++k) {
#pragma unroll
                for (int q = 0; q < NP; ++q) ms[k][q] = mrow[(uint64_t)k * B4 + q * 256];
            }
            auto step = [&](uint32_t st, int k) { // k = st % MD
                const uint32_t bk = st + MD < WL_BUCKETS ? st + MD : WL_BUCKETS - 1;
#pragma unroll
                for (int q = 0; q < NP; ++q) ms[k][q] = mrow[(uint64_t)bk * B4 + q * 256];
                if (!DB) lrb_barrier(); // everybody is through with this bucket of the map
                wl_v4u *mdb = md + (DB ? ((st + 1) & 1u) * B4 : 0);
#pragma unroll
                for (int q = 0; q < NP; ++q) mdb[q * 256] = ms[(k + 1) % MD][q];
                lrb_barrier();
            };
            uint32_t i = 0;
            for (; i + MD <= WL_BUCKETS; i += MD) {
#pragma unroll
                for (int k = 0; k < MD; ++k) step(i + k, k);
            }
#pragma unroll
            for (int k = 0; k < (int)(WL_BUCKETS % MD); ++k) step(i + k, k);
        } else {
            // ---- entries: lane rt takes entries b0 + rt + ET m (m < NE) of a bucket
            const uint32_t rt = (wave - LW) * 64 + lane;
            uint32_t ring[WL_SWEEP_DEPTH][NE];
            // bv: lane l holds bounds[8 blk + l] of the current block of eight steps (lanes 0..16 are used: the
            // steps' own bounds and those of the eight steps after, whose entries are asked for); asked for a block ahead
            auto ask_bounds = [&](uint32_t blk) {
                const uint32_t i = blk * 8 + lane;
                return bg[i < WL_BUCKETS ? i : WL_BUCKETS];
            };
            auto refill = [&](uint32_t c0, uint32_t c1, uint32_t *e) {
                const __amdgpu_buffer_rsrc_t rs =
                    __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(lg), 0, (int)(c1 * 4u), 0x00020000);
                const uint32_t v = (c0 + rt) * 4u;
#pragma unroll
                for (uint32_t m = 0; m < NE; ++m) e[m] = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(v + m * ET * 4u), 0, 0);
            };
            auto tally = [&](uint32_t e, uint32_t bin) {
                atomicAdd(&hist[__umul24(bin, Rh) + (e >> (WL_SLICE_BITS + 1))], 1u << ((e >> (WL_SLICE_BITS - 4)) & 16u)); // (bin < 256, Rh <= 1,016: the 24-bit multiply is a full-rate instruction, the 32-bit one a quarter-rate)
            };
            uint32_t bv = ask_bounds(0), bv_next = ask_bounds(1);
#pragma unroll
            for (int k = 0; k < WL_SWEEP_DEPTH; ++k)
                refill((uint32_t)__builtin_amdgcn_readlane(bv, k), (uint32_t)__builtin_amdgcn_readlane(bv, k + 1), ring[k]);
            for (uint32_t blk = 0; blk < WL_BUCKETS / 8; ++blk) {
                const uint32_t bv_after = ask_bounds(blk + 2);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane(bv, k), b1 = (uint32_t)__builtin_amdgcn_readlane(bv, k + 1);
                    const uint32_t cnt = b1 - b0;
                    const uint8_t *mb = map_s + (DB ? (k & 1) * BB : 0); // (eight steps a block: step parity = k parity)
                    auto bin_of = [&](uint32_t e) -> uint32_t {
                        const uint32_t idx = e & 0x7FFFu;
                        if (BITS == 8) return mb[idx];
                        // PER pairs to a word: word idx / PER (by a multiply for PER = 6: exact below 2^15), field idx % PER
                        constexpr uint32_t PER = 32 / BITS;
                        const uint32_t w = PER == 8 ? idx >> 3 : (idx * 10923u) >> 16;
                        return (reinterpret_cast<const uint32_t *>(mb)[w] >> ((idx - w * PER) * BITS)) & ((1u << BITS) - 1u);
                    };
                    // the map bytes of all the step's entries are read before any is tallied: one LDS round trip
                    uint32_t bin[NE];
#pragma unroll
                    for (uint32_t m = 0; m < NE; ++m)
                        if (cnt > m * ET) bin[m] = bin_of(ring[k][m]);
#pragma unroll
                    for (uint32_t m = 0; m < NE; ++m)
                        if (cnt > m * ET && rt + m * ET < cnt) tally(ring[k][m], bin[m]);
                    if (cnt > NE * ET) // (a bucket longer than the ring covers: repeats, low-complexity reads)
                        for (uint32_t q = b0 + rt + NE * ET; q < b1; q += ET) {
                            const uint32_t e = lg[q];
                            tally(e, bin_of(e));
                        }
                    refill((uint32_t)__builtin_amdgcn_readlane(bv, k + 8), (uint32_t)__builtin_amdgcn_readlane(bv, k + 9), ring[k]);
                    lrb_barrier();
                    if (!DB) lrb_barrier();
                }
                bv = bv_next;
                bv_next = bv_after;
            }
        }
        const uint32_t nr = (uint32_t)(r1 - r0);
        uint32_t *ho = hist_out + r0 * bins;
        for (uint32_t i = tid; i < nr * bins; i += NT) {
            const uint32_t r = i / bins, b = i - r * bins;
            ho[i] = (hist[b * Rh + (r >> 1)] >> ((r & 1u) << 4)) & 0xFFFFu;
        }
        for (uint32_t r = tid; r < nr; r += NT) {
            uint32_t sum = 0;
            for (uint32_t b = 0; b < bins; ++b) sum += (hist[b * Rh + (r >> 1)] >> ((r & 1u) << 4)) & 0xFFFFu;
            sums_out[r0 + r] = sum;
        }
    }
}

// ===========================================================================
// host side
// ===========================================================================
// reads per group: at most what the LDS holds of u16 counters beside the map bucket, in WHOLE rounds of one workgroup
// per CU (a round of the sweep costs the same 16,384 steps whatever the group size)
uint64_t lrb_wl_group_reads(const lrb_ctx *c, uint64_t n, int bins, uint64_t total_bases)
{
    // (the sweep keeps two reads' u16 counters in one word: an EVEN number of reads must fit -- lrb_wl_hist_fits)
    uint64_t rmax = (WL_HIST_CAP / (uint32_t)bins) & ~1u;
    if (rmax > WL_MAX_READS) rmax = WL_MAX_READS;
    // ... and a group's (group, slice) lists should fit the order kernel's registers (WL_ORDER_CACHE x 1024 entries;
    // longer ones are streamed twice -- correct, slower): 62,500 windows a slice on average when the lengths are known
    if (total_bases && n) {
        const uint64_t per_read = total_bases / n > 14 ? total_bases / n - 14 : 1;
        const uint64_t cap = 62500ull * WL_SLICES / per_read;
        if (cap >= 64 && cap < rmax) rmax = cap;
    }
    if (rmax < 1) rmax = 1;
    const uint64_t slots = (uint64_t)c->n_cu;
    const uint64_t rounds = (n + slots * rmax - 1) / (slots * rmax);
    uint64_t R = (n + slots * rounds - 1) / (slots * (rounds ? rounds : 1));
    if (R < 64) R = 64;
    if (const char *e = getenv("LRB_K3_SWEEP_READS")) R = strtoull(e, nullptr, 10); // experiments, tests
    uint64_t hard = ((WL_HIST_CAP / (uint32_t)bins) & ~1u) < WL_MAX_READS ? ((WL_HIST_CAP / (uint32_t)bins) & ~1u) : WL_MAX_READS;
    if (hard < 1) hard = 1;
    if (R > hard) R = hard;
    if (R < 1) R = 1;
    return R;
}

// can lists cut for groups of R reads be swept for histograms of `bins` bins?  (the one predicate of the sweep, of
// lrb_winlists_cov_hist and of PackedLists.fits in device.py)
bool lrb_wl_hist_fits(uint64_t R, int bins)
{
    return bins >= 1 && bins <= 256 && R >= 1 && R <= WL_MAX_READS && ((R + 1) & ~1ull) * (uint64_t)bins <= WL_HIST_CAP;
}

static uint32_t wl_units(uint64_t R) { return R >= 256 ? 4u : R >= 64 ? 2u : 1u; }

extern "C" int lrb_k15_lists_geometry_for(lrb_ctx *c, uint64_t n, uint64_t total_bases, int bins, uint32_t *reads_per_group,
                                          uint64_t *n_groups)
{
    ARG_TRY(c != nullptr && reads_per_group != nullptr && n_groups != nullptr);
    ARG_TRY(bins >= 1 && bins <= 256);
    const uint64_t R = lrb_wl_group_reads(c, n ? n : 1, bins, total_bases);
    *reads_per_group = (uint32_t)R;
    *n_groups = (n + R - 1) / R;
    return LRB_OK;
}

extern "C" int lrb_k15_lists_geometry(lrb_ctx *c, uint64_t n, int bins, uint32_t *reads_per_group, uint64_t *n_groups)
{
    ARG_TRY(c != nullptr && reads_per_group != nullptr && n_groups != nullptr);
    ARG_TRY(bins >= 1 && bins <= 256);
    const uint64_t R = lrb_wl_group_reads(c, n ? n : 1, bins, 0);
    *reads_per_group = (uint32_t)R;
    *n_groups = (n + R - 1) / R;
    return LRB_OK;
}

extern "C" uint64_t lrb_k15_lists_bounds_words(uint64_t n_groups) { return n_groups * (uint64_t)WL_BSTRIDE; }

// The level-1 scratch (slot 9) sized for lists of `list_slots` entries BEFORE the first partition of a series: the part
// call keeps whatever size it finds from 2 GB on and goes through larger lists in chunks of that, so a series that starts
// with a small set of reads (lrb_packed_k15_tally_half_many_for fills its groups from the end: the FIRST group is the
// remainder) would partition every later, full group in three or four chunks -- measured: 47 ms of part kernel per 2.5 M
// reads instead of 30.  Same budget rule as the part call (a quarter of the free memory, at most 24 GB).
int lrb_wl_reserve_scratch(lrb_ctx *c, uint64_t list_slots)
{
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    uint64_t budget = (uint64_t)free_b / 4 + c->ws_bytes[9];
    if (budget > (24ull << 30)) budget = 24ull << 30;
    uint64_t want = list_slots * sizeof(uint32_t) + 64;
    if (want > budget) want = budget;
    if (c->ws_bytes[9] >= want) return LRB_OK;
    void *p;
    return lrb_ws_get(c, 9, want, &p);
}

extern "C" int lrb_k15_lists_part_dev(lrb_ctx *c, const uint32_t *d_codes, const uint32_t *d_mask,
                                      const uint64_t *d_code_off, const uint64_t *d_mask_off, const uint32_t *d_lens,
                                      uint64_t n, uint32_t reads_per_group, uint32_t *d_lists, uint32_t *d_bounds,
                                      uint64_t *d_gbase)
{
    ARG_TRY(c != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    if (n == 0) return LRB_OK;
    ARG_TRY(d_codes && d_mask && d_code_off && d_mask_off && d_lens && d_lists && d_bounds && d_gbase);
    ARG_TRY(reads_per_group >= 1 && reads_per_group <= WL_MAX_READS);
    const uint64_t ngroups64 = (n + reads_per_group - 1) / reads_per_group;
    ARG_TRY(ngroups64 <= 0x7FFFFFFFull / WL_MAX_UNITS);
    const uint32_t ngroups = (uint32_t)ngroups64, R = reads_per_group;
    const uint32_t P = wl_units(R);
    const uint32_t Ru = (R + P - 1) / P;
    ARG_TRY(Ru <= WLR_READS); // (the part kernel's read table)
    hipLaunchKernelGGL(wl_gbase_kernel, dim3((ngroups + 256) / 256), dim3(256), 0, c->stream, d_mask_off, n, R, ngroups,
                       d_gbase);
    std::vector<uint64_t> gb(ngroups + 1);
    HIP_TRY(hipMemcpyAsync(gb.data(), d_gbase, sizeof(uint64_t) * (ngroups + 1), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    // the level-1 lists of a CHUNK of groups live in scratch (slot 9): what it holds already when that is 2 GB or
    // more, else a quarter of the free memory, at most 24 GB; a chunk is whole rounds of the part kernel's grid
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    uint64_t budget = (uint64_t)free_b / 4 + c->ws_bytes[9];
    if (budget > (24ull << 30)) budget = 24ull << 30;
    // (kept as it is from 2 GB on -- unless these lists are more than twice that: a series that began with a small set of
    // reads then grows the scratch once instead of cutting every later set into chunks)
    const uint64_t need = (gb[ngroups] - gb[0]) * sizeof(uint32_t) + 64;
    if ((c->ws_bytes[9] >= (2ull << 30) && need <= 2 * c->ws_bytes[9]) || budget < c->ws_bytes[9]) budget = c->ws_bytes[9];
    if (const char *e = getenv("LRB_K3_SWEEP_WS_MB")) budget = strtoull(e, nullptr, 10) << 20; // tests
    const uint64_t budget_slots = budget / 4;
    uint32_t order_run = 8;
    if (const char *e = getenv("LRB_WL_ORDER_RUN")) order_run = (uint32_t)strtoul(e, nullptr, 10); // experiments
    if (order_run < 1) order_run = 1;
    void *d_small, *d_flag;
    {
        const int rc = lrb_ws_get(c, 18, 64, &d_flag);
        if (rc != LRB_OK) return rc;
    }
    // LRB_WL_FAULT_AT=k (tests): the k-th partition of the process finds one of its counts one short on its first attempt
    static const uint64_t fault_at = getenv("LRB_WL_FAULT_AT") ? strtoull(getenv("LRB_WL_FAULT_AT"), nullptr, 10) : 0;
    static uint64_t part_calls_total = 0;
    const uint64_t part_calls = ++part_calls_total;
    for (int attempt = 0;; ++attempt) {
    HIP_TRY(hipMemsetAsync(d_flag, 0, 4, c->stream));
    uint32_t g0 = 0;
    while (g0 < ngroups) {
        uint32_t g1 = g0 + 1; // one group at least (its scratch is allocated whatever the budget says)
        while (g1 < ngroups && gb[g1 + 1] - gb[g0] <= budget_slots) ++g1;
        if (g1 < ngroups && g1 - g0 > (uint32_t)c->n_cu) g1 = g0 + (g1 - g0) / c->n_cu * c->n_cu;
        const uint32_t gc = g1 - g0, nunits = gc * P;
        void *d_tmp;
        int rc = lrb_ws_get(c, 9, (gb[g1] - gb[g0]) * sizeof(uint32_t) + 64, &d_tmp);
        if (rc != LRB_OK) return rc;
        rc = lrb_ws_get(c, 10, (uint64_t)nunits * WL_SLICES * sizeof(uint32_t) * 2 + 64, &d_small);
        if (rc != LRB_OK) return rc;
        uint32_t *d_cnt1 = (uint32_t *)d_small, *d_start1 = d_cnt1 + (uint64_t)nunits * WL_SLICES;
        const unsigned gw = (unsigned)(nunits < 4u * c->n_cu ? nunits : 4u * c->n_cu);
        hipLaunchKernelGGL(wl_count_kernel, dim3(gw), dim3(1024), 0, c->stream, d_codes, d_mask, d_code_off, d_mask_off,
                           d_lens, n, R, Ru, P, g0, nunits, d_cnt1);
        if (fault_at && attempt == 0 && g0 == 0 && part_calls == fault_at)   // (tests: one count made one short, once)
            hipLaunchKernelGGL(wl_fault_kernel, dim3(1), dim3(1), 0, c->stream, d_cnt1);
        hipLaunchKernelGGL(wl_gscan_kernel, dim3(gc), dim3(256), 0, c->stream, (const uint32_t *)d_cnt1, P, g0, d_start1,
                           d_bounds);
        {
            static lrb_per_device_once ring_attr;
            if (ring_attr.need(c->device))
                HIP_TRY(hipFuncSetAttribute((const void *)wl_part_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WLR_SMEM_BYTES));
            // (one workgroup a CU: the rings take 128 of its 160 KB)
            const unsigned gr = (unsigned)(nunits < (uint32_t)c->n_cu ? nunits : (uint32_t)c->n_cu);
            hipLaunchKernelGGL(wl_part_kernel, dim3(gr), dim3(1024), WLR_SMEM_BYTES, c->stream, d_codes, d_mask, d_code_off,
                               d_mask_off, d_lens, n, R, Ru, P, g0, nunits, (const uint64_t *)d_gbase, (uint32_t *)d_tmp,
                               (const uint32_t *)d_start1, (const uint32_t *)d_cnt1, (uint32_t *)d_flag);
        }
        for (uint32_t gy = 0; gy < gc; gy += 32768) {
            const uint32_t ny = gc - gy < 32768 ? gc - gy : 32768;
            // (the scratch is addressed from the chunk's first group: tmp shifted so that group g0 + gy reads its own)
            const uint32_t *tmp_y = (const uint32_t *)d_tmp + (gb[g0 + gy] - gb[g0]);
            // (occ1: a workgroup orders the slice's lists of up to order_run groups in turn, each list asked for under
            // the one before it)
            // and the streamed kernel behind it for the lists of more than 65,536 entries (it looks at every list's length
            // and leaves at once where there is none: some 20 us)
            const dim3 grid_run(WL_SLICES, (ny + order_run - 1) / order_run);
            hipLaunchKernelGGL(wl_order_kernel_occ1, grid_run, dim3(1024), 0, c->stream, tmp_y, g0 + gy, ny,
                               (const uint64_t *)d_gbase, d_lists, d_bounds, (const uint32_t *)d_bounds);
            hipLaunchKernelGGL(wl_order_kernel, grid_run, dim3(1024), 0, c->stream, tmp_y, g0 + gy, ny,
                               (uint32_t)(WL_ORDER_CACHE * 1024), (const uint64_t *)d_gbase, d_lists, d_bounds,
                               (const uint32_t *)d_bounds);
        }
        HIP_TRY(hipGetLastError());
        g0 = g1;
    }
    // count and part walked the same windows?  (one word back and a wait for the stream: the callers' next kernels are
    // enqueued behind it, some tens of microseconds a partition of milliseconds)
    uint32_t mismatch = 0;
    HIP_TRY(hipMemcpyAsync(&mismatch, d_flag, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (!mismatch) break;
    ++c->wl_retries;
    if (attempt >= 4) {
        lrb_set_error("window lists: the count and the part kernel disagree after %s%s", "five attempts", "");
        return LRB_ERR_HIP;
    }
    }
    return LRB_OK;
}

extern "C" int lrb_k15_lists_tally_dev(lrb_ctx *c, const uint32_t *d_codes, const uint32_t *d_mask,
                                       const uint64_t *d_code_off, const uint64_t *d_mask_off, const uint32_t *d_lens,
                                       uint64_t n, uint32_t reads_per_group, const uint32_t *d_lists,
                                       const uint32_t *d_bounds, const uint64_t *d_gbase, uint32_t *d_half)
{
    ARG_TRY(c != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    if (n == 0) return LRB_OK;
    ARG_TRY(d_codes && d_mask && d_code_off && d_mask_off && d_lens && d_lists && d_bounds && d_gbase && d_half);
    ARG_TRY(reads_per_group >= 1 && reads_per_group <= WL_MAX_READS);
    const uint64_t ngroups = (n + reads_per_group - 1) / reads_per_group;
    ARG_TRY(ngroups <= 0x7FFFFFFFull / WL_MAX_UNITS);
    static lrb_per_device_once attr_done;
    if (attr_done.need(c->device)) {
        HIP_TRY(hipFuncSetAttribute((const void *)wl_tally_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    }
    hipLaunchKernelGGL(wl_tally_kernel, dim3(WL_BUCKETS), dim3(1024), 131072, c->stream, d_lists, d_bounds, d_gbase,
                       (uint32_t)ngroups, d_half);
    HIP_TRY(hipGetLastError());
    // the reads the lists leave out (more than 65,535 windows): one atomic per window
    return lrb_k15_accum_half_long(c, d_codes, d_mask, d_code_off, d_mask_off, d_lens, n, WL_MAX_WINDOWS + 15u, d_half);
}

// The sweep in the form chosen (two buckets of the map in LDS or one; 8 / 5 / 4 bits a pair): twelve entry waves and two
// register sets a loader wave -- eight and three for the byte map with two buckets in LDS, where twelve would need 52 bytes
// a lane of scratch (round 4's A/B: 4 entry waves 11.5 ms, 8: 8.5, 12: 8.2 per 4e9 windows).
template <bool DBV, int BITS, int NP>
static int wl_sweep_launch(lrb_ctx *c, unsigned grid, size_t smem, const uint32_t *d_lists, const uint32_t *d_bounds,
                           const uint64_t *d_gbase, uint64_t n, uint32_t reads_per_group, uint64_t ngroups, const void *d_use,
                           int bins, uint32_t *d_hist, uint32_t *d_sums)
{
    constexpr int EW = (DBV && BITS == 8) ? 8 : 12, MD = EW == 8 ? 3 : 2;
    static lrb_per_device_once once_;
    if (once_.need(c->device))
        HIP_TRY(hipFuncSetAttribute((const void *)wl_sweep_kernel<4, EW, MD, DBV, BITS, NP>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    163840));
    hipLaunchKernelGGL((wl_sweep_kernel<4, EW, MD, DBV, BITS, NP>), dim3(grid), dim3(64 * (4 + EW)), smem, c->stream, d_lists, d_bounds,
                       d_gbase, n, reads_per_group, (uint32_t)ngroups, (const uint8_t *)d_use, (uint32_t)bins, d_hist, d_sums);
    return LRB_OK;
}

extern "C" int lrb_cov_lists_sweep_dev(lrb_ctx *c, const uint32_t *d_codes, const uint32_t *d_mask,
                                       const uint64_t *d_code_off, const uint64_t *d_mask_off, const uint32_t *d_lens,
                                       uint64_t n, uint32_t reads_per_group, const uint32_t *d_lists,
                                       const uint32_t *d_bounds, const uint64_t *d_gbase, const uint8_t *d_map, int bins,
                                       uint32_t *d_hist, uint32_t *d_sums)
{
    ARG_TRY(c != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    ARG_TRY(bins >= 1 && bins <= 256);
    if (n == 0) return LRB_OK;
    ARG_TRY(d_codes && d_mask && d_code_off && d_mask_off && d_lens && d_lists && d_bounds && d_gbase && d_map && d_hist && d_sums);
    ARG_TRY(lrb_wl_hist_fits(reads_per_group, bins));
    const uint64_t ngroups = (n + reads_per_group - 1) / reads_per_group;
    ARG_TRY(ngroups <= 0x7FFFFFFFull / WL_MAX_UNITS);
    const size_t hbytes = ((size_t)((reads_per_group + 1) / 2) * bins * 4 + 15) & ~(size_t)15;
    const unsigned grid = (unsigned)(ngroups < (uint64_t)c->n_cu ? ngroups : (uint64_t)c->n_cu);
    // the form of the map: a byte a pair as it is handed in (8 pieces of 4 KB a bucket), or packed here to 5 / 4 bits a pair
    // (6 / 4 pieces) when the histogram has at most 32 / 16 bins.  Two buckets of the map in LDS when the histograms leave
    // room for them (one barrier a step)
    const int form = bins <= 16 ? 4 : bins <= 32 ? 5 : 8;
    const size_t bb = form == 8 ? 32768 : form == 5 ? 24576 : 16384;
    const bool db = 2 * bb + hbytes <= 163840;
    const void *d_use = d_map;
    if (form != 8) {
        void *d_packed;
        int rc = lrb_ws_get(c, 16, (uint64_t)WL_BUCKETS * bb + 64, &d_packed);
        if (rc != LRB_OK) return rc;
        if (form == 5)
            hipLaunchKernelGGL((wl_map_pack_kernel<5, 6>), dim3(c->n_cu * 16), dim3(256), 0, c->stream, d_map, (uint32_t *)d_packed);
        else
            hipLaunchKernelGGL((wl_map_pack_kernel<4, 4>), dim3(c->n_cu * 16), dim3(256), 0, c->stream, d_map, (uint32_t *)d_packed);
        d_use = d_packed;
    }
    const size_t smem = (db ? 2 : 1) * bb + hbytes;
    int rc_l;
    if (form == 8) rc_l = db ? wl_sweep_launch<true, 8, 8>(c, grid, smem, d_lists, d_bounds, d_gbase, n, reads_per_group, ngroups, d_use, bins, d_hist, d_sums)
                             : wl_sweep_launch<false, 8, 8>(c, grid, smem, d_lists, d_bounds, d_gbase, n, reads_per_group, ngroups, d_use, bins, d_hist, d_sums);
    else if (form == 5) rc_l = db ? wl_sweep_launch<true, 5, 6>(c, grid, smem, d_lists, d_bounds, d_gbase, n, reads_per_group, ngroups, d_use, bins, d_hist, d_sums)
                                  : wl_sweep_launch<false, 5, 6>(c, grid, smem, d_lists, d_bounds, d_gbase, n, reads_per_group, ngroups, d_use, bins, d_hist, d_sums);
    else rc_l = db ? wl_sweep_launch<true, 4, 4>(c, grid, smem, d_lists, d_bounds, d_gbase, n, reads_per_group, ngroups, d_use, bins, d_hist, d_sums)
                   : wl_sweep_launch<false, 4, 4>(c, grid, smem, d_lists, d_bounds, d_gbase, n, reads_per_group, ngroups, d_use, bins, d_hist, d_sums);
    if (rc_l != LRB_OK) return rc_l;
    HIP_TRY(hipGetLastError());
    return lrb_cov_hist_map_long(c, d_codes, d_mask, d_code_off, d_mask_off, d_lens, n, d_map, bins, d_hist, d_sums);
}

// ---- K3 of reads nobody has cut into lists yet: window lists in the context's workspaces, then the sweep ----
// K3 of one range of reads whose mask words (`words` of them) fit the workspace: window lists (slot 8; their level-1
// scratch is slot 9), bounds and group bases (slot 11), then the sweep
static int cov_sweep_range(lrb_ctx *c, const uint32_t *d_codes, const uint32_t *d_mask, const uint64_t *d_code_off,
                           const uint64_t *d_mask_off, const uint32_t *d_lens, uint64_t n, uint64_t words,
                           const uint8_t *d_map, int bins, uint32_t *d_hist, uint32_t *d_sums)
{
    const uint64_t R = lrb_wl_group_reads(c, n, bins, words * 32); // (32 base slots a mask word: the padded lengths)
    const uint64_t ngroups = (n + R - 1) / R;
    void *d_buf, *d_small;
    int rc = lrb_ws_get(c, 8, words * 32 * sizeof(uint32_t) + 64, &d_buf);
    if (rc != LRB_OK) return rc;
    const uint64_t bwords = lrb_k15_lists_bounds_words(ngroups);
    rc = lrb_ws_get(c, 11, bwords * sizeof(uint32_t) + (ngroups + 1) * sizeof(uint64_t) + 64, &d_small);
    if (rc != LRB_OK) return rc;
    uint64_t *d_gbase = (uint64_t *)d_small;
    uint32_t *d_bounds = (uint32_t *)(d_gbase + ngroups + 1);
    rc = lrb_k15_lists_part_dev(c, d_codes, d_mask, d_code_off, d_mask_off, d_lens, n, (uint32_t)R, (uint32_t *)d_buf,
                                d_bounds, d_gbase);
    if (rc != LRB_OK) return rc;
    return lrb_cov_lists_sweep_dev(c, d_codes, d_mask, d_code_off, d_mask_off, d_lens, n, (uint32_t)R,
                                   (const uint32_t *)d_buf, d_bounds, d_gbase, d_map, bins, d_hist, d_sums);
}

// Where to cut a batch whose slice lists would not fit the workspace: out[0] = number of ranges, then per range
// {end read, mask_off[end]}.  A range takes reads while their mask words stay within the budget (one read at least).
__global__ void cov_sweep_splits_kernel(const uint64_t *__restrict__ mask_off, uint64_t n, uint64_t budget_words,
                                        uint64_t *__restrict__ out, uint64_t max_ranges)
{
    if (threadIdx.x | blockIdx.x) return;
    uint64_t r = 0, k = 0;
    while (r < n && k < max_ranges) {
        const uint64_t base = mask_off[r];
        uint64_t lo = r + 1, hi = n; // the end is in [lo, hi]; mask_off[lo] counts as fitting
        if (mask_off[hi] - base <= budget_words) {
            lo = hi;
        } else {
            while (hi - lo > 1) { // mask_off[hi] - base > budget
                const uint64_t mid = (lo + hi) >> 1;
                if (mask_off[mid] - base <= budget_words) lo = mid;
                else hi = mid;
            }
        }
        out[1 + 2 * k] = lo;
        out[2 + 2 * k] = mask_off[lo];
        ++k;
        r = lo;
    }
    out[0] = r < n ? ~0ull : k; // ~0: more ranges than the caller provided for
}

extern "C" int lrb_cov_hist_sweep_dev(lrb_ctx *c, const uint32_t *d_codes, const uint32_t *d_mask,
                                      const uint64_t *d_code_off, const uint64_t *d_mask_off,
                                      const uint32_t *d_lens, uint64_t n, const uint8_t *d_map, int bins,
                                      uint32_t *d_hist, uint32_t *d_sums)
{
    ARG_TRY(c != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    ARG_TRY(bins >= 1 && bins <= 256);
    if (n == 0) return LRB_OK;
    ARG_TRY(d_codes && d_mask && d_code_off && d_mask_off && d_lens && d_map && d_hist && d_sums);
    // the slice lists take 32 slots per mask word of the batch: read back where the batch's mask words end
    uint64_t ends[2];
    HIP_TRY(hipMemcpyAsync(&ends[0], d_mask_off, sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&ends[1], d_mask_off + n, sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    ARG_TRY(ends[1] >= ends[0]);
    const uint64_t words = ends[1] - ends[0];
    // workspace budget: what slot 8 holds already when that is 4 GB or more (growing it costs 25 ms of hipMalloc per
    // GB, sweeping a batch in a few ranges costs next to nothing), else half of the free memory, at most 24 GB
    // (4.7e9 bases a range)
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    uint64_t budget = (uint64_t)free_b / 3 + c->ws_bytes[8]; // (the level-1 scratch of the part kernel wants as much again)
    if (budget > (24ull << 30)) budget = 24ull << 30;
    if (c->ws_bytes[8] >= (4ull << 30) || budget < c->ws_bytes[8]) budget = c->ws_bytes[8];
    if (const char *e = getenv("LRB_K3_SWEEP_WS_MB")) budget = strtoull(e, nullptr, 10) << 20; // tests
    uint64_t budget_words = budget / 128;
    if (budget_words < 4096) budget_words = 4096;
    int rc = LRB_OK;
    if (words <= budget_words) {
        rc = cov_sweep_range(c, d_codes, d_mask, d_code_off, d_mask_off, d_lens, n, words, d_map, bins, d_hist, d_sums);
    } else {
        // every range but the last holds more than half the budget unless single reads are larger than that
        const uint64_t max_ranges = 2 * (words / budget_words) + 8 < n ? 2 * (words / budget_words) + 8 : n;
        void *d_spl;
        rc = lrb_ws_get(c, 10, (1 + 2 * max_ranges) * sizeof(uint64_t), &d_spl);
        if (rc != LRB_OK) return rc;
        hipLaunchKernelGGL(cov_sweep_splits_kernel, dim3(1), dim3(64), 0, c->stream, d_mask_off, n, budget_words,
                           (uint64_t *)d_spl, max_ranges);
        std::vector<uint64_t> spl(1 + 2 * max_ranges);
        HIP_TRY(hipMemcpyAsync(spl.data(), d_spl, spl.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (spl[0] == ~0ull) {
            lrb_set_error("coverage sweep: reads too long for the workspace budget%s%s", "", "");
            return LRB_ERR_NOMEM;
        }
        uint64_t r = 0, w = ends[0];
        for (uint64_t k = 0; k < spl[0] && rc == LRB_OK; ++k) {
            const uint64_t e = spl[1 + 2 * k], we = spl[2 + 2 * k];
            rc = cov_sweep_range(c, d_codes, d_mask, d_code_off + r, d_mask_off + r, d_lens + r, e - r, we - w, d_map, bins,
                                 d_hist + r * bins, d_sums + r);
            r = e;
            w = we;
        }
    }
    return rc; // (reads of more than 65,535 windows: the gather kernel, inside lrb_cov_lists_sweep_dev)
}

