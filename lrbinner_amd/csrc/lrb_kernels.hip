// lrb_kernels.hip -- gfx950 (MI355X / CDNA4) kernels + the device half of the C ABI
// declared in include/lrb_hip.h.  Written for wave64 / 160 KB LDS / HBM3E only.
//
// Kernels (DESIGN.md has the roofline and byte accounting of each):
//   pack_kernel        ASCII -> 2-bit codes + ACGT validity mask
//   k1_count_kernel    canonical k-mer tallies per read      (count-kmers.cpp:66-87)
//   k15_accum_kernel   F[val] += 1 over valid 15-mers        (kmer_utils.h:114-156)
//   k15_mirror_kernel  T[x] = F[x] + F[rc(x)]                (kmer_utils.h:146-153)
//   cov_hist_kernel    gather T[val], bin, tally per read    (kmer_utils.h:24-72)
//   seed_dist_kernel   0.5 - M @ M[seed]                     (cluster_utils.py:45-49)
//   seed_hist_kernel   histc(distances, 60, 0, 0.3) x seeds  (cluster_utils.py:137-139)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <fcntl.h>
#include <unistd.h>

#include <new>
#include <string>
#include <thread>
#include <mutex>
#include <vector>

#include "lrb_device.h"
#include "lrb_k15_dev.h"

// ---------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------
// Order this wave's LDS traffic across lanes.  DS operations of one wave execute in
// issue order, so only the compiler has to be kept from moving them.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// 4 ASCII bytes (first base in the lowest byte) -> 8 bits of codes, first base in bits 7..6.
// (w>>1)&3 per byte, then one multiply gathers the four 2-bit fields.
__device__ __forceinline__ uint32_t code8_of(uint32_t w)
{
    uint32_t t = (w >> 1) & 0x03030303u;
    return (t * 0x40100401u) >> 24;
}

// 4 ASCII bytes -> 4 validity bits (first base in bit 3): byte is exactly 'A','C','G','T'.
// The byte's own 2-bit code selects the letter it would have to be (v_perm_b32 table
// lookup); equal bytes are valid.
__device__ __forceinline__ uint32_t valid4_of(uint32_t w)
{
    uint32_t t = (w >> 1) & 0x03030303u;
    uint32_t expect = __builtin_amdgcn_perm(0u, 0x47544341u /* 'A','C','T','G' */, t);
    uint32_t diff = w ^ expect;
    uint32_t nz = (((diff & 0x7f7f7f7fu) + 0x7f7f7f7fu) | diff) & 0x80808080u;
    uint32_t u = (~nz & 0x80808080u) >> 7;
    return ((u * 0x08040201u) >> 24) & 0xFu;
}

// bit 0 of each of 4 bytes -> 4 bits, first byte in bit 3
__device__ __forceinline__ uint32_t bit4_of(uint32_t w)
{
    return (((w & 0x01010101u) * 0x08040201u) >> 24) & 0xFu;
}

// ---------------------------------------------------------------------------
// pack: one wave per read (grid-stride), one lane per 32-base chunk.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_kernel(const uint8_t *__restrict__ seqs,
                                                   const uint64_t *__restrict__ offs, uint64_t n,
                                                   const uint64_t *__restrict__ code_off,
                                                   const uint64_t *__restrict__ mask_off,
                                                   uint32_t *__restrict__ codes,
                                                   uint32_t *__restrict__ mask,
                                                   uint32_t *__restrict__ planes)
{
    const uint32_t lane = lane_id();
    const uint64_t wave0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = wave0; r < n; r += nwaves) {
        const uint64_t b = offs[r];
        const uint64_t L = offs[r + 1] - b;
        uint32_t *cw = codes + code_off[r];
        const uint64_t ncw = code_off[r + 1] - code_off[r]; // multiple of 4
        uint32_t *mw = mask ? mask + mask_off[r] : nullptr;
        uint2 *pw = planes ? reinterpret_cast<uint2 *>(planes + 2 * mask_off[r]) : nullptr;
        const uint64_t nmw = (mask || planes) ? mask_off[r + 1] - mask_off[r] : 0;
        const uint64_t nchunks = (ncw >> 1) > nmw ? (ncw >> 1) : nmw;
        const uint8_t *p0 = seqs + b;
        for (uint64_t c = lane; c < nchunks; c += WAVE) {
            const uint64_t base = c << 5;
            uint32_t c0 = 0, c1 = 0, m = 0, ph = 0, pl = 0;
            if (base + 32 <= L) {
                uint32_t d[8];
                __builtin_memcpy(d, p0 + base, 32);
                c0 = (code8_of(d[0]) << 24) | (code8_of(d[1]) << 16) | (code8_of(d[2]) << 8) |
                     code8_of(d[3]);
                c1 = (code8_of(d[4]) << 24) | (code8_of(d[5]) << 16) | (code8_of(d[6]) << 8) |
                     code8_of(d[7]);
                if (mw) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) m |= valid4_of(d[j]) << (28 - 4 * j);
                }
                if (pw) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        ph |= bit4_of(d[j] >> 2) << (28 - 4 * j);
                        pl |= bit4_of(d[j] >> 1) << (28 - 4 * j);
                    }
                }
            } else if (base < L) {
                const uint32_t rem = (uint32_t)(L - base);
                for (uint32_t i = 0; i < rem; ++i) {
                    const uint32_t ch = p0[base + i];
                    const uint32_t code = (ch >> 1) & 3u;
                    if (i < 16)
                        c0 |= code << (30 - 2 * i);
                    else
                        c1 |= code << (30 - 2 * (i - 16));
                    const uint32_t ok = (ch == 'A') | (ch == 'C') | (ch == 'G') | (ch == 'T');
                    m |= ok << (31 - i);
                    ph |= (code >> 1) << (31 - i);
                    pl |= (code & 1u) << (31 - i);
                }
            }
            if ((c << 1) < ncw) {
                uint2 v;
                v.x = c0;
                v.y = c1;
                *reinterpret_cast<uint2 *>(cw + (c << 1)) = v;
            }
            if (mw && c < nmw) mw[c] = m;
            if (pw && c < nmw) {
                uint2 v;
                v.x = ph;
                v.y = pl;
                pw[c] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// K1: canonical k-mer tallies.  One wave per read (4 independent waves per
// workgroup, grid-stride).  Per wave an LDS histogram of 4^K bins x SUBS u32
// sub-counters laid out [bin][sub]: lane l always hits bank (bin*SUBS + l%SUBS)%32,
// so with SUBS=16 at most two lanes of a 32-lane group share a bank, which the
// ds_add_u32 data path absorbs (scripts/ubench_lds.hip: same rate as SUBS=32).
// ds_add_u32 issues at ~4 cycles per wave-instruction per CU; that, not HBM, is
// what bounds this kernel.  Each lane walks two runs of 16 consecutive
// bases (a word and its halo word each) per trip; the next trip's words -- or the
// next read's first words -- are in flight while the current ones are tallied.
// ---------------------------------------------------------------------------
typedef __attribute__((address_space(3))) uint32_t lds_u32_t;

__device__ __forceinline__ uint32_t lds_addr_of(const void *p)
{
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}

__device__ __forceinline__ void lds_inc(uint32_t byte_addr)
{
    __hip_atomic_fetch_add((lds_u32_t *)(uintptr_t)byte_addr, 1u, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
}

// LDS byte address of the sub-counter for the k-mer starting at base p (0..15) of word
// hi (lo = following word): two VALU ops (shift/alignbit, and-or).
template <int K, int SH>
__device__ __forceinline__ uint32_t tally_addr(uint32_t hi, uint32_t lo, int p, uint32_t laneoff)
{
    constexpr uint32_t FM = ((1u << (2 * K)) - 1u) << SH; // k-mer field, pre-scaled
    const int used = 2 * p + 2 * K;
    uint32_t t;
    if (used + SH <= 32)
        t = hi >> (32 - used - SH);
    else
        t = __builtin_amdgcn_alignbit(hi, lo, 64 - used - SH);
    return (t & FM) | laneoff;
}

// One lane's share of a trip: its 16-base word and the following (halo) word.
struct k1_words {
    uint32_t w, h;
};

// Loaded through a buffer resource over the read's words (+ the pad word the region carries past its last word):
// past the end the range check returns zeros, so there is no branch around the load -- a predicated load is a
// branch, and at its join the compiler waits for everything in flight.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t k1_rsrc(const uint32_t *cw, uint32_t ncw)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(cw), 0, ncw ? (int)((ncw + 1) * 4u) : 0, 0x00020000);
}

__device__ __forceinline__ k1_words k1_bload(__amdgpu_buffer_rsrc_t rs, uint32_t wi, uint32_t ncw)
{
    k1_words r;
    r.w = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(wi * 4u), 0, 0);
    r.h = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(wi * 4u), 4, 0); // word ncw is the region's pad word
    (void)ncw; // words at or past ncw hold no window start that is tallied (the callers mask by position)
    return r;
}

// The LDS-histogram path for one wave: reads r0, r0+stride, ... < n.  A trip is TW x 64 words: every lane
// holds TW word pairs and tallies 16 TW windows while the next trip's loads are in flight -- with one pair per
// lane (16 ds_add) a wave ran out of work long before its prefetch came back, and 16 waves per CU kept the LDS
// pipe only half busy.
template <int K, int SUBS, int TW = 2>
__device__ __forceinline__ void k1_lds_loop(const uint32_t *__restrict__ codes,
                                            const uint64_t *__restrict__ code_off,
                                            const uint32_t *__restrict__ lens, uint64_t r0,
                                            uint64_t stride, uint64_t n, uint32_t *hist,
                                            uint32_t *canon, const uint16_t *lut_s, uint32_t dim,
                                            uint32_t *__restrict__ counts, uint32_t lane)
{
    constexpr int BINS = 1 << (2 * K);
    constexpr int HWORDS = BINS * SUBS;
    constexpr int SH = (SUBS == 32 ? 7 : SUBS == 16 ? 6 : SUBS == 8 ? 5 : 4); // log2(SUBS*4)
    static_assert(SUBS == 32 || SUBS == 16 || SUBS == 8 || SUBS == 4, "SUBS");
    const uint32_t laneoff = lds_addr_of(hist) + (lane & (SUBS - 1)) * 4u;
    uint64_t r = r0;
    if (r >= n) return;

    constexpr uint32_t TRIP = WAVE * TW;
    uint32_t L = lens[r];
    const uint32_t *cw = codes + code_off[r];
    k1_words cur[TW];
    {
        const __amdgpu_buffer_rsrc_t rs = k1_rsrc(cw, (L + 15) >> 4);
#pragma unroll
        for (int u = 0; u < TW; ++u) cur[u] = k1_bload(rs, lane + u * WAVE, (L + 15) >> 4);
    }

    for (;;) {
        // metadata of this wave's next read: in flight during the whole tally
        const uint64_t rn = r + stride;
        const bool has_next = rn < n;
        uint32_t Ln = 0;
        uint64_t offn = 0;
        if (has_next) {
            Ln = lens[rn];
            offn = code_off[rn];
        }
        // clear this wave's histogram
        {
            const uint4 z = {0u, 0u, 0u, 0u};
            uint4 *h4 = reinterpret_cast<uint4 *>(hist);
#pragma unroll
            for (int i = 0; i < HWORDS / 4 / WAVE; ++i) h4[i * WAVE + lane] = z;
            for (uint32_t i = lane; i < dim; i += WAVE) canon[i] = 0;
        }
        wave_lds_fence();

        const uint32_t nk = L >= (uint32_t)K ? L - K + 1 : 0; // window start positions
        const uint32_t ncw = (L + 15) >> 4;
        const uint32_t ncwn = has_next ? (Ln + 15) >> 4 : 0;
        const __amdgpu_buffer_rsrc_t rs_cur = k1_rsrc(cw, ncw), rs_next = k1_rsrc(codes + offn, ncwn);
        // one trip = TW x 64 words: lane l owns words it + l, it + 64 + l, ... (coalesced 256-B loads)
        for (uint32_t it = 0;; it += TRIP) {
            const bool last = it + TRIP >= ncw; // wave-uniform
            k1_words nxt[TW];
            if (!last) {
#pragma unroll
                for (int u = 0; u < TW; ++u) nxt[u] = k1_bload(rs_cur, it + TRIP + u * WAVE + lane, ncw);
            } else {
#pragma unroll
                for (int u = 0; u < TW; ++u) nxt[u] = k1_bload(rs_next, u * WAVE + lane, ncwn);
            }
            if (((uint64_t)it + TRIP) * 16 <= nk) { // wave-uniform: every window is real
#pragma unroll
                for (int u = 0; u < TW; ++u)
#pragma unroll
                    for (int p = 0; p < 16; ++p) lds_inc(tally_addr<K, SH>(cur[u].w, cur[u].h, p, laneoff));
            } else {
#pragma unroll
                for (int u = 0; u < TW; ++u) {
                    const uint32_t pos0 = (it + u * WAVE + lane) * 16;
#pragma unroll
                    for (int p = 0; p < 16; ++p) {
                        const uint32_t a = tally_addr<K, SH>(cur[u].w, cur[u].h, p, laneoff);
                        if (pos0 + p < nk) lds_inc(a);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < TW; ++u) cur[u] = nxt[u];
            if (last) break;
        }
        wave_lds_fence();

        // fold the sub-counters of each bin (16-B reads, start rotated per lane so the 16
        // lanes served together fall in 16 different bank groups), map through the
        // canonical LUT, then one coalesced row store
        for (int b = lane; b < BINS; b += WAVE) {
            uint32_t s = 0;
            if (SUBS >= 4) {
                const uint4 *hb = reinterpret_cast<const uint4 *>(hist + b * SUBS);
#pragma unroll
                for (int q = 0; q < SUBS / 4; ++q) {
                    const uint4 v = hb[(q + (lane >> 2)) & (SUBS / 4 - 1)];
                    s += v.x + v.y + v.z + v.w;
                }
            }
            atomicAdd(&canon[lut_s[b]], s);
        }
        wave_lds_fence();
        uint32_t *out = counts + r * dim;
        for (uint32_t i = lane; i < dim; i += WAVE) out[i] = canon[i];
        wave_lds_fence();

        if (!has_next) break;
        r = rn;
        L = Ln;
        cw = codes + offn;
    }
}

template <int K, int SUBS, int WAVES_PER_SIMD, int TW = 2>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void k1_count_kernel(
    const uint32_t *__restrict__ codes, const uint64_t *__restrict__ code_off,
    const uint32_t *__restrict__ lens, uint64_t n, const uint16_t *__restrict__ lut,
    uint32_t dim, uint32_t dimpad, uint32_t *__restrict__ counts)
{
    constexpr int BINS = 1 << (2 * K);
    constexpr int HWORDS = BINS * SUBS;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t lane = lane_id();
    uint32_t *hist = smem + wave * HWORDS;
    uint32_t *canon = smem + 4 * HWORDS + wave * dimpad;
    uint16_t *lut_s = reinterpret_cast<uint16_t *>(smem + 4 * HWORDS + 4 * dimpad);
    for (int i = threadIdx.x; i < BINS; i += 256) lut_s[i] = lut[i];
    lrb_barrier();
    k1_lds_loop<K, SUBS, TW>(codes, code_off, lens, (uint64_t)blockIdx.x * 4 + wave,
                         (uint64_t)gridDim.x * 4, n, hist, canon, lut_s, dim, counts, lane);
}

// ---------------------------------------------------------------------------
// K1, k = 3, bit-plane form: no LDS atomics at all.  The read is held as two bit
// planes per 32-base block (H = high bit of every 2-bit code, L = low bit, first base
// in bit 31).  Integer VALU issues about one wave-instruction per clock per CU on
// gfx950, so the cost of this path is its VALU instruction count; the formulation
// below needs 88 per 32 bases (2.75 per base):
//
//   x = (a, b, c) and rc(x) = (c', b', a') (x' = x ^ 2) differ in the HIGH bit of the
//   middle base and share its LOW bit.  With P[a][c] = [base0 = a][base2 = c]
//   (16 masks, one 3-input op each) the union of the two strands' windows with middle
//   high bit bh is ONE mux,  w = H1 == bh ? P[a][c] : P[c'][a'],  and the low middle bit
//   splits it:  class(b_l = 1) = popcount(w & L1),  class(b_l = 0) = popcount(w) -
//   popcount(w & L1).  That is 16 groups x {mux, and, 2 popcount-accumulate}; the
//   subtraction happens once per read.  k is odd, so the two strands never share a
//   window and the union is disjoint.
//
// Per read the 32 accumulators are summed across the wave with a halving butterfly on
// the cross-lane paths that cost no LDS round trip (v_permlane32/16_swap, DPP row_ror:8,
// row_half_mirror, quad_perm); lane l ends up holding slot (l >> 1).
// ---------------------------------------------------------------------------
struct k3_groups {
    unsigned char a[16], c[16], bh[16], same[16]; // representative (b_l = 0) of each group
    unsigned char slot_class[32];                 // canonical class of slot g / 16+g
};

constexpr k3_groups make_k3_groups()
{
    k3_groups t = {};
    int lut[64] = {};
    int next = 0;
    for (int x = 0; x < 64; ++x) {
        // reverse complement of a 3-mer code (count-kmers.cpp:24-36): reverse groups, XOR 2
        const int rc = (((x & 3) ^ 2) << 4) | ((((x >> 2) & 3) ^ 2) << 2) | (((x >> 4) & 3) ^ 2);
        lut[x] = rc < x ? lut[rc] : next++;
    }
    int g = 0;
    for (int x = 0; x < 64; ++x) {
        const int a = x >> 4, b = (x >> 2) & 3, c = x & 3;
        const int rc = ((c ^ 2) << 4) | ((b ^ 2) << 2) | (a ^ 2);
        if ((b & 1) != 0 || rc < x) continue; // one representative per group: b_l = 0, x <= rc(x)
        t.a[g] = (unsigned char)a;
        t.c[g] = (unsigned char)c;
        t.bh[g] = (unsigned char)(b >> 1);
        t.same[g] = (unsigned char)(a == (c ^ 2)); // P[a][c] and P[c'][a'] coincide
        t.slot_class[g] = (unsigned char)lut[x];
        t.slot_class[16 + g] = (unsigned char)lut[x | 4]; // same window with b_l = 1
        ++g;
    }
    return t;
}


typedef unsigned v2u_t __attribute__((ext_vector_type(2)));

// Keeps the 32 running tallies where they are between blocks: without it the compiler
// re-associates the sums of unrolled blocks into popcount + add3 chains (one more vector
// instruction per tally) instead of v_bcnt's built-in accumulate.
__device__ __forceinline__ void acc_pin(uint32_t (&acc)[32])
{
#pragma unroll
    for (int q = 0; q < 32; ++q) asm("" : "+v"(acc[q]));
}

// Tally the 3-mers that START in this 32-base block.  (Hn, Ln) = next block (halo);
// V = 1 bits for start positions that exist (all ones in the interior of the read).
//
// SKIP_LAST: group k3_skip_group() is not tallied.  Every window is in exactly one group,
// so its two classes follow per read from the window count and from the number of windows
// with b_l = 1, which is kept in acc[16 + skip] (one popcount instead of mux + and + two).
constexpr int k3_skip_group()
{
    constexpr k3_groups T = make_k3_groups();
    int g = 15;
    while (T.same[g]) --g;
    return g;
}

template <bool SKIP_LAST = false>
__device__ __forceinline__ void swar3_block(uint32_t H, uint32_t L, uint32_t Hn, uint32_t Ln,
                                            uint32_t V, uint32_t (&acc)[32])
{
    constexpr k3_groups T = make_k3_groups();
    constexpr int GS = k3_skip_group();
    const uint32_t H1 = __builtin_amdgcn_alignbit(H, Hn, 31), L1 = __builtin_amdgcn_alignbit(L, Ln, 31);
    const uint32_t H2 = __builtin_amdgcn_alignbit(H, Hn, 30), L2 = __builtin_amdgcn_alignbit(L, Ln, 30);
    uint32_t e0[4];
    e0[0] = ~H & ~L & V; // A
    e0[1] = ~H & L & V;  // C
    e0[2] = H & ~L & V;  // T
    e0[3] = H & L & V;   // G
    uint32_t P[16];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        P[a * 4 + 0] = e0[a] & ~H2 & ~L2;
        P[a * 4 + 1] = e0[a] & ~H2 & L2;
        P[a * 4 + 2] = e0[a] & H2 & ~L2;
        P[a * 4 + 3] = e0[a] & H2 & L2;
    }
    // keep the compiler from re-deriving the 16 masks inside every use (it would trade
    // the shared mux for three 3-input ops per accumulator)
#pragma unroll
    for (int i = 0; i < 16; ++i) asm("" : "+v"(P[i]));
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        if (SKIP_LAST && g == GS) {
            acc[16 + g] += __builtin_popcount(L1 & V);
            continue;
        }
        const uint32_t fw = P[T.a[g] * 4 + T.c[g]];
        uint32_t w = fw;
        if (!T.same[g]) {
            const uint32_t rv = P[(T.c[g] ^ 2) * 4 + (T.a[g] ^ 2)];
            w = T.bh[g] ? ((H1 & fw) | (~H1 & rv)) : ((~H1 & fw) | (H1 & rv));
        }
        asm("" : "+v"(w));
        acc[g] += __builtin_popcount(w);
        acc[16 + g] += __builtin_popcount(w & L1);
    }
}


// ---------------------------------------------------------------------------
// K1, k = 3, LANE-PER-READ form of the bit-plane kernel.  The wave-per-read kernel above
// spends 40 % of its instructions outside the 84-per-block core: 86 to set a read up, 92
// to fold 32 accumulators across 64 lanes, 13 per trip.  Here a wave owns a GROUP of 64
// reads and lane l walks read 64g+l block by block: its 32 accumulators are that read's
// tallies (no fold), set-up and loop control are shared by 64 reads.  For the loads to
// coalesce the bit planes are kept group-transposed: block j of the 64 reads of group g
// is one contiguous 512-B row,  planes_t[(group_off[g] + j) * 64 + l]  (uint2 {H, L}),
// group_off[g+1]-group_off[g] = 1 + max blocks of the group (the extra row is the zero
// halo).  Reads of different length in one group cost the longest one's time.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k1_swar3_lane_kernel(const uint2 *__restrict__ planes_t,
                                                            const uint64_t *__restrict__ group_off,
                                                            const uint32_t *__restrict__ order,
                                                            const uint32_t *__restrict__ lens,
                                                            uint64_t n,
                                                            uint32_t *__restrict__ counts)
{
    constexpr k3_groups T = make_k3_groups();
    const uint32_t lane = lane_id();
    const uint64_t ngroups = (n + 63) >> 6;
    // everything that steers the loops is wave-uniform and kept in scalar registers, so
    // the block loop spends its vector issue slots on the tally alone
    const uint32_t wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t wave0 = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wave_in_block;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint32_t voff = lane * 8u;
    for (uint64_t g = wave0; g < ngroups; g += nwaves) {
        const uint64_t slot = (g << 6) + lane;
        const bool have = slot < n;
        const uint64_t r = have ? (order ? order[slot] : slot) : 0; // the read in this slot
        const uint32_t L = have ? lens[r] : 0u;
        const uint32_t nk = L >= 3 ? L - 2 : 0;
        const uint64_t row0 = group_off[g];
        const uint32_t rows = (uint32_t)(group_off[g + 1] - row0); // 1 + max blocks
        // windows every lane of the wave still has in full blocks (wave-uniform)
        uint32_t nk_min = nk;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t other = __shfl_xor(nk_min, o, WAVE);
            nk_min = other < nk_min ? other : nk_min;
        }
        nk_min = __builtin_amdgcn_readfirstlane(nk_min);
        const uint32_t last = rows - 1; // the last row of the group (halo of the last block)
        const uint32_t full = nk_min / 32 < last ? nk_min / 32 : last; // blocks with V = all ones
        // rows come in through a buffer descriptor rebuilt (scalar ALU) at the current row:
        // per-lane offset in a VGPR, row offsets as immediates, rows past the group's end
        // read as zero by the range check
        const char *base = reinterpret_cast<const char *>(planes_t) + row0 * 512;
        auto rsrc_at = [&](uint32_t j) {
            const uint64_t left = j < rows ? (uint64_t)(rows - j) * 512 : 0; // past the end: zero records, loads return 0
            return __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char *>(base + (uint64_t)j * 512), 0,
                left < 0x7FFFFFFFull ? (int)left : 0x7FFFFFFF, 0x00020000);
        };
        auto load_row = [&](__amdgpu_buffer_rsrc_t rs, int imm) -> uint2 {
            const v2u_t v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff, imm, 0);
            return make_uint2(v.x, v.y);
        };
        uint32_t acc[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) acc[q] = 0;
        // three rows rotate through ra/rb/rc without register moves: row j+2 is in flight
        // while row j (with its halo j+1) is tallied
        uint2 ra, rb, rc;
        {
            const auto rs = rsrc_at(0);
            ra = load_row(rs, 0);
            rb = load_row(rs, 512);
            rc = load_row(rs, 1024);
        }
        uint32_t j = 0;
        for (; j + 3 <= full; j += 3) {
            const auto rs = rsrc_at(j + 3);
            swar3_block<true>(ra.x, ra.y, rb.x, rb.y, 0xFFFFFFFFu, acc);
            ra = load_row(rs, 0);
            acc_pin(acc);
            swar3_block<true>(rb.x, rb.y, rc.x, rc.y, 0xFFFFFFFFu, acc);
            rb = load_row(rs, 512);
            acc_pin(acc);
            swar3_block<true>(rc.x, rc.y, ra.x, ra.y, 0xFFFFFFFFu, acc);
            rc = load_row(rs, 1024);
            acc_pin(acc);
        }
        // the ragged end of the group: per-lane validity of the window starts
        for (; j < last; ++j) {
            const uint32_t p0 = j * 32;
            uint32_t V = 0;
            if (p0 + 32 <= nk)
                V = 0xFFFFFFFFu;
            else if (p0 < nk)
                V = 0xFFFFFFFFu << (32 - (nk - p0));
            swar3_block<true>(ra.x, ra.y, rb.x, rb.y, V, acc);
            ra = rb;
            rb = rc;
            rc = load_row(rsrc_at(j + 3), 0);
        }
        {
            // the group left out of the tally: what the others leave of the totals
            constexpr int GS = k3_skip_group();
            uint32_t s0 = 0, s1 = 0;
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (q != GS) {
                    s0 += acc[q];
                    s1 += acc[16 + q];
                }
            acc[GS] = nk - s0;
            acc[16 + GS] -= s1;
        }
        if (have) {
            uint32_t *out = counts + r * 32;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                out[T.slot_class[q]] = acc[q] - acc[16 + q]; // b_l = 0 class
                out[T.slot_class[16 + q]] = acc[16 + q];     // b_l = 1 class
            }
        }
    }
}

// planes (per read) -> planes_t (group-transposed) through a 64-read x 32-block LDS tile:
// 256-B runs of one read in, 512-B rows of one block out.
__global__ __launch_bounds__(256) void planes_t_kernel(const uint32_t *__restrict__ planes,
                                                       const uint64_t *__restrict__ mask_off,
                                                       const uint64_t *__restrict__ group_off,
                                                       const uint32_t *__restrict__ order,
                                                       uint64_t n, uint2 *__restrict__ planes_t)
{
    __shared__ uint2 tile[64][33];
    const uint64_t g = blockIdx.x;
    const uint64_t row0 = group_off[g];
    const uint32_t rows = (uint32_t)(group_off[g + 1] - row0);
    const uint32_t t = threadIdx.x;
    for (uint32_t j0 = 0; j0 < rows; j0 += 32) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t rr = (t >> 5) + 8 * i, b = t & 31;
            const uint64_t slot = (g << 6) + rr;
            uint2 v = {0u, 0u};
            if (slot < n) {
                const uint64_t r = order ? order[slot] : slot;
                const uint64_t nb = mask_off[r + 1] - mask_off[r]; // blocks in this read's region
                if (j0 + b < nb)
                    v = reinterpret_cast<const uint2 *>(planes + 2 * mask_off[r])[j0 + b];
            }
            tile[rr][b] = v;
        }
        lrb_barrier();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t b = (t >> 6) + 4 * i, l = t & 63;
            if (j0 + b < rows) planes_t[(row0 + j0 + b) * 64 + l] = tile[l][b];
        }
        lrb_barrier();
    }
}

// ASCII -> group-transposed bit planes directly (what the host path uses for k = 3).
// One workgroup per group of 64 reads; a tile of 32 blocks x 64 reads goes through LDS:
// lanes read 32 consecutive bytes of one read (two reads per wave at a time), rows of
// 512 B go out.
__global__ __launch_bounds__(256) void pack_planes_t_kernel(const uint8_t *__restrict__ seqs,
                                                            const uint64_t *__restrict__ offs,
                                                            const uint64_t *__restrict__ group_off,
                                                            const uint32_t *__restrict__ order,
                                                            uint64_t n, uint2 *__restrict__ planes_t)
{
    __shared__ uint2 tile[64][33];
    const uint64_t g = blockIdx.x;
    const uint64_t row0 = group_off[g];
    const uint32_t rows = (uint32_t)(group_off[g + 1] - row0);
    const uint32_t t = threadIdx.x;
    for (uint32_t j0 = 0; j0 < rows; j0 += 32) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t rr = (t >> 5) + 8 * i, b = t & 31;
            const uint64_t slot = (g << 6) + rr;
            uint32_t ph = 0, pl = 0;
            if (slot < n) {
                const uint64_t r = order ? order[slot] : slot;
                const uint64_t beg = offs[r], L = offs[r + 1] - beg;
                const uint64_t base = (uint64_t)(j0 + b) << 5;
                const uint8_t *p = seqs + beg + base;
                if (base + 32 <= L) {
                    uint32_t d[8];
                    __builtin_memcpy(d, p, 32);
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        ph |= bit4_of(d[q] >> 2) << (28 - 4 * q);
                        pl |= bit4_of(d[q] >> 1) << (28 - 4 * q);
                    }
                } else if (base < L) {
                    const uint32_t rem = (uint32_t)(L - base);
                    for (uint32_t q = 0; q < rem; ++q) {
                        const uint32_t code = ((uint32_t)p[q] >> 1) & 3u;
                        ph |= (code >> 1) << (31 - q);
                        pl |= (code & 1u) << (31 - q);
                    }
                }
            }
            tile[rr][b] = make_uint2(ph, pl);
        }
        lrb_barrier();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t b = (t >> 6) + 4 * i, l = t & 63;
            if (j0 + b < rows) planes_t[(row0 + j0 + b) * 64 + l] = tile[l][b];
        }
        lrb_barrier();
    }
}

// ---------------------------------------------------------------------------
// K1, k = 4, LANE-PER-READ with private histograms in LDS.  The wave-per-read LDS kernel
// (k1_count_kernel) spends 60 % of its LDS-array cycles on bank conflicts (rocprofv3:
// SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, profiles/r02_k1_k4_lds_rocprof_summary.txt): 64
// lanes of ONE read scatter over [bin][8 sub-counters].  Here a wave owns a GROUP of 64 reads
// and lane l tallies read order[64g+l] into ITS OWN column of the histogram,
//     hist[bin][lane & 31] : uint32 = {u16 of lane l | u16 of lane l+32},  256 bins x 128 B = 32 KB,
// so the bank is the lane: no two lanes of a 32-lane group ever meet, no tally is ever shared
// (ds_add_u32 of 1 or 65536), nothing is folded across lanes and the histogram is cleared once
// per 64 reads instead of once per read.  What is left is the cost of moving address + data of
// one ds_add_u32 per window to the LDS: 4 cycles per wave-instruction per CU.
// The 2-bit codes are kept group-transposed like the k = 3 planes: row j of group g holds code
// words 4j..4j+3 (64 bases) of its 64 reads, codes_t[(group_off[g] + j) * 64 + l] (uint4), one
// contiguous 1-KiB line per load; group_off[g+1] - group_off[g] = 1 + max rows (zero halo row).
// A u16 column counter holds 65535: a group is tallied in chunks of at most 1023 rows
// (65472 windows), each flushed to the output (stored by the first chunk, added by later ones).
// Canonical classes exactly as compute_kmer_inds (count-kmers.cpp:38-64) numbers them.
// ---------------------------------------------------------------------------
template <int K> struct kmer_classes {
    static constexpr int BINS = 1 << (2 * K);
    unsigned short fw[BINS / 2 + 16], rc[BINS / 2 + 16]; // 136 / 512 classes are in use
    int n;
};

template <int K> constexpr kmer_classes<K> make_kmer_classes()
{
    kmer_classes<K> t = {};
    int next = 0;
    for (int x = 0; x < (1 << (2 * K)); ++x) {
        int rc = 0;
        for (int i = 0; i < K; ++i) rc |= (((x >> (2 * i)) & 3) ^ 2) << (2 * (K - 1 - i));
        if (rc < x) continue; // its class was opened by rc
        t.fw[next] = (unsigned short)x;
        t.rc[next] = (unsigned short)rc;
        ++next;
    }
    t.n = next;
    return t;
}

__device__ __forceinline__ void lds_add(uint32_t byte_addr, uint32_t v)
{
    __hip_atomic_fetch_add((lds_u32_t *)(uintptr_t)byte_addr, v, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
}

// the 64 windows that start in one row (w0..w3) of this lane's read; halo = first word of the
// next row.  PRED: only windows whose start is below nk (the ragged end of a group).
template <int K, int SH, bool PRED>
__device__ __forceinline__ void lane4_row(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t halo,
                                          uint32_t laneoff, uint32_t one, uint32_t pos0, uint32_t nk)
{
    const uint32_t w[5] = {w0, w1, w2, w3, halo};
    // With one or two waves per SIMD nothing hides the latency between an instruction and the one that
    // uses its result, so a word's 16 windows go through the three steps side by side: 16 shifts, 16
    // mask-and-merge, 16 ds_add (the scheduling barriers keep the compiler from chaining them again).
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t t[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const int used = 2 * p + 2 * K;
            t[p] = used + SH <= 32 ? w[q] >> (32 - used - SH) : __builtin_amdgcn_alignbit(w[q], w[q + 1], 64 - used - SH);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < 16; ++p) t[p] = (t[p] & (((1u << (2 * K)) - 1u) << SH)) | laneoff;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < 16; ++p)
            if (!PRED || pos0 + (uint32_t)(q * 16 + p) < nk) lds_add(t[p], one);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---------------------------------------------------------------------------
// Flush of a HALF-group histogram (32 columns, two to a word, 16 words = 64 B per bin), class-major (round 3).
// The first form of the flush gave every read's column to 16 threads, each walking its share of the classes behind
// 34 (k = 4) or 128 (k = 5) exec-masked blocks with half the lanes idle and one ds_read_u16 per bin and column:
// 5.6 us per half group, as long as the tally it follows (scripts/k4s2_probe.hip).  Here a thread owns a CLASS and
// EIGHT columns: the bins of the class are read as 16-byte rows (ds_read_b128: four words = eight columns), summed
// as packed u16 pairs (a column's tallies add up to at most 65,280 per chunk, so no half ever carries), unpacked
// and stored -- 16 consecutive classes of one read by 16 lanes, 64-byte runs.  Lane l of a wave takes column
// group l & 3 and class 16 * (pass * W + wave) + (l >> 2); its four prefix bins are read in an order rotated by
// (l >> 2) & 3, so the 16 lanes of a ds_read_b128 service group touch 16 different bank quads.
//   S2 (k = 4 from 5-mers at even positions): class c <- bins 4 x + b and 256 a + x for x in {fw, rc}, plus the
//       closing 4-mer of an even-length read (tail[column], 0xFFFFFFFF: none);
//   else (k = 5): class c <- bins fw, rc.
// cls[c] = fw | rc << 16; rcol[column] = output row of the column's read, ~0 when the column is empty.
// ---------------------------------------------------------------------------
template <int DIM, bool S2, int W, int KPOW = 256>   // KPOW = 4^k (S2: the bins are (k+1)-mers)
__device__ __forceinline__ void lane4_flush_half(const uint32_t *smem, const uint32_t *tail, const uint64_t *rcol,
                                                 const uint32_t *cls, uint32_t *__restrict__ counts, uint32_t lane,
                                                 uint32_t wv, bool add_prev, bool with_tail)
{
    const uint32_t cg = lane & 3u, ci = lane >> 2;
    const uint4 *rows = reinterpret_cast<const uint4 *>(smem) + cg; // bin b: rows[4 * b]
    auto add4 = [](uint4 &a, const uint4 v) { a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; };
    constexpr int PASSES = (DIM + 16 * W - 1) / (16 * W);
#pragma unroll 1
    for (int pass = 0; pass < PASSES; ++pass) {
        const uint32_t c = 16u * ((uint32_t)pass * W + wv) + ci;
        if (c >= (uint32_t)DIM) continue;
        const uint32_t fr = cls[c];
        const uint32_t fw = fr & 0xFFFFu, rc = fr >> 16;
        uint4 acc = {0u, 0u, 0u, 0u};
        if (S2) {
            const uint32_t rot = ci & 3u;
#pragma unroll
            for (int b = 0; b < 4; ++b) add4(acc, rows[4u * (4u * fw + ((b + rot) & 3u))]);
#pragma unroll
            for (int a = 0; a < 4; ++a) add4(acc, rows[4u * ((uint32_t)KPOW * a + fw)]);
            if (rc != fw) {
#pragma unroll
                for (int b = 0; b < 4; ++b) add4(acc, rows[4u * (4u * rc + ((b + rot) & 3u))]);
#pragma unroll
                for (int a = 0; a < 4; ++a) add4(acc, rows[4u * ((uint32_t)KPOW * a + rc)]);
            }
        } else {
            acc = rows[4u * fw];
            if (rc != fw) add4(acc, rows[4u * rc]);
        }
        uint32_t v[8] = {acc.x & 0xFFFFu, acc.x >> 16, acc.y & 0xFFFFu, acc.y >> 16,
                         acc.z & 0xFFFFu, acc.z >> 16, acc.w & 0xFFFFu, acc.w >> 16};
        if (S2 && with_tail) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t t = tail[8u * cg + j];
                v[j] += (t == fw || t == rc) ? 1u : 0u;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint64_t r = rcol[8u * cg + j];
            if (r == ~0ull) continue;
            uint32_t *p = counts + r * DIM + c;
            if (add_prev) v[j] += *p;
            *p = v[j];
        }
    }
}

// W waves of a workgroup share the 64 columns: wave w tallies rows w, w+W, ... of the group (the
// atomics make that safe, and two waves never meet in one LDS cycle).  A wave that has just issued a
// ds_add cannot issue anything else for ~16 cycles (operand transfer to the LDS), so the 2 vector
// instructions per window of ONE wave never overlap its own tallies: it takes several waves per SIMD
// to keep the LDS path busy (scripts/ubench_lds.hip: 3.4 / 2.3 / 2.0 / 1.8 ns per ds_add at 4 / 8 / 16 /
// 32 waves per CU), and 32 KB per group would allow only five.

// HALF: a workgroup owns HALF a group -- 32 reads, 32 columns (two to a word), lanes l and l+32 taking
// the even and the odd row of a row pair of read l -- so that a histogram is half as large and two
// workgroups share a CU: one's flush overlaps the other's tally (k = 5: 2 x 64 KB instead of 128 KB).
template <int K, int W, int NR, bool HALF>
__global__ __launch_bounds__(64 * W) void k1_lane4_kernel(const uint4 *__restrict__ codes_t,
                                                          const uint64_t *__restrict__ group_off,
                                                          const uint32_t *__restrict__ order,
                                                          const uint32_t *__restrict__ lens, uint64_t n,
                                                          uint32_t *__restrict__ counts)
{
    constexpr kmer_classes<K> T = make_kmer_classes<K>();
    constexpr int DIM = T.n;                  // 136 / 512
    constexpr int U = HALF ? 2 : 1;           // rows a wave takes per step (one per lane)
    constexpr int SH = HALF ? 6 : 7;          // log2 of a bin's bytes: 32 or 16 words
    constexpr int CLR = (1 << (2 * K)) / (HALF ? 16 : 8); // 1-KiB slabs (64 lanes x 16 B) of the histogram
    // steps between flushes: a column takes 64 U windows per step and holds 65535; multiple of W * NR
    constexpr uint32_t K4_CHUNK = (1020 / U / (W * NR)) * (W * NR);
    static_assert(DIM % 4 == 0 && (K == 4 || K == 5) && K4_CHUNK > 0, "K");
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[]; // 4^K bins x (32 | 16) words
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t g = HALF ? blockIdx.x >> 1 : blockIdx.x;
    const uint32_t col = HALF ? (lane & 31u) : lane;                 // the read's column in the histogram
    const uint32_t sub = HALF ? (lane >> 5) : 0u;                    // which row of the step's row pair
    const uint32_t in_group = HALF ? (blockIdx.x & 1u) * 32u + col : lane;
    const uint64_t slot = (g << 6) + in_group;
    const bool have = slot < n;
    const uint64_t r = have ? (order ? order[slot] : slot) : 0;
    const uint32_t L = have ? lens[r] : 0u;
    const uint32_t nk = L >= (uint32_t)K ? L - (K - 1) : 0;
    const uint64_t row0 = group_off[g];
    const uint32_t rows = (uint32_t)(group_off[g + 1] - row0); // 1 + max rows
    const uint32_t last = rows - 1;
    uint32_t nk_min = nk;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t other = __shfl_xor(nk_min, o, WAVE);
        nk_min = other < nk_min ? other : nk_min;
    }
    nk_min = __builtin_amdgcn_readfirstlane(nk_min);
    const uint32_t full = nk_min / 64 < last ? nk_min / 64 : last; // rows whose 64 windows exist in every lane
    const uint32_t ufull = full / U;                                // steps made of such rows only
    const uint32_t ulast = (last + U - 1) / U;                      // steps that cover every row below the halo row

    // column c: word c (lanes c and c + 32 share it, low / high half) or, HALF, word c / 2 (columns 2w, 2w + 1)
    const uint32_t laneoff = lds_addr_of(smem) + (HALF ? (col >> 1) : (lane & 31u)) * 4u;
    const uint32_t one = (HALF ? (col & 1u) : (lane >> 5)) ? 0x10000u : 1u;
    const char *base = reinterpret_cast<const char *>(codes_t) + row0 * 1024;
    const uint32_t voff = sub * 1024u + in_group * 16u;
    auto rsrc_at = [&](uint32_t j) {
        const uint64_t left = j < rows ? (uint64_t)(rows - j) * 1024 : 0; // past the end: loads return 0
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base + (uint64_t)j * 1024), 0,
                                                 left < 0x7FFFFFFFull ? (int)left : 0x7FFFFFFF, 0x00020000);
    };
    typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
    struct row_t {
        v4u_t w;
        uint32_t halo; // first word of the following row (another lane's or another wave's row)
    };
    auto load_row = [&](__amdgpu_buffer_rsrc_t rs, int imm) -> row_t {
        row_t x;
        x.w = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, imm, 0);
        x.halo = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, imm + 1024, 0);
        return x;
    };
    auto clear = [&]() {
        const uint4 z = {0u, 0u, 0u, 0u};
        uint4 *h4 = reinterpret_cast<uint4 *>(smem);
#pragma unroll
        for (int i = 0; i < CLR / W; ++i) h4[(i * W + wv) * 64 + lane] = z;
        if (CLR % W != 0 && (CLR / W) * W + wv < CLR) h4[((CLR / W) * W + wv) * 64 + lane] = z;
        lrb_barrier();
    };
    // my steps: q(m) = wv + W * m, rows U q .. U q + U - 1
    auto mine_below = [&](uint32_t bound) { return bound > wv ? (bound - wv + W - 1) / W : 0u; };

    row_t R[NR];
    {
        const auto rs = rsrc_at(U * wv);
#pragma unroll
        for (int i = 0; i < NR; ++i) R[i] = load_row(rs, i * W * U * 1024);
    }
    if (HALF) {   // tables of the class-major flush, behind the histogram: output row per column, fw | rc << 16 per class
        uint32_t *xtra = smem + (1 << (2 * K)) * 16;
        if (threadIdx.x < 32) reinterpret_cast<uint64_t *>(xtra)[col] = have ? r : ~0ull;
        for (uint32_t c = threadIdx.x; c < (uint32_t)DIM; c += 64 * W) xtra[64 + c] = (uint32_t)T.fw[c] | ((uint32_t)T.rc[c] << 16);
    }
    clear();
    const uint32_t mfull = mine_below(ufull);
    uint32_t m = 0;
    for (uint32_t c0 = 0;; c0 += K4_CHUNK) {
        const uint32_t c1 = c0 + K4_CHUNK < ulast ? c0 + K4_CHUNK : ulast; // this chunk: steps c0 .. c1
        const uint32_t mc1 = mine_below(c1);
        const uint32_t fast = mfull < mc1 ? mfull : mc1;
        for (; m + NR <= fast; m += NR) {
            const auto rs = rsrc_at(U * (wv + W * (m + NR)));
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                lane4_row<K, SH, false>(R[i].w.x, R[i].w.y, R[i].w.z, R[i].w.w, R[i].halo, laneoff, one, 0u, 0u);
                R[i] = load_row(rs, i * W * U * 1024);
            }
        }
        for (; m < mc1; ++m) {
            const uint32_t q = wv + W * m;
            if (q < ufull)
                lane4_row<K, SH, false>(R[0].w.x, R[0].w.y, R[0].w.z, R[0].w.w, R[0].halo, laneoff, one, 0u, 0u);
            else
                lane4_row<K, SH, true>(R[0].w.x, R[0].w.y, R[0].w.z, R[0].w.w, R[0].halo, laneoff, one,
                                       (U * q + sub) * 64u, nk);
#pragma unroll
            for (int i = 0; i + 1 < NR; ++i) R[i] = R[i + 1];
            R[NR - 1] = load_row(rsrc_at(U * (wv + W * (m + NR))), 0);
        }
        // flush: this read's column -> its canonical tallies, four to a store, the stores dealt round the
        // waves (and, HALF, the two lanes of a read)
        lrb_barrier();
        if constexpr (HALF) {
            // class-major flush (lane4_flush_half): a thread owns a class and eight columns
            const uint32_t *xtra = smem + (1 << (2 * K)) * 16;
            lane4_flush_half<DIM, false, W>(smem, nullptr, reinterpret_cast<const uint64_t *>(xtra), xtra + 64, counts,
                                            lane, wv, c0 != 0, false);
        } else if (have) {
            const unsigned char *colp = reinterpret_cast<const unsigned char *>(smem) +
                                        (HALF ? (col >> 1) * 4u + (col & 1u) * 2u : (lane & 31u) * 4u + (lane >> 5) * 2u);
            uint4 *out = reinterpret_cast<uint4 *>(counts + r * DIM);
            const uint32_t me = wv * U + sub;
#pragma unroll
            for (int c4 = 0; c4 < DIM / 4; ++c4) {
                if ((uint32_t)(c4 % (W * U)) != me) continue;
                uint32_t v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = c4 * 4 + e;
                    v[e] = *reinterpret_cast<const uint16_t *>(colp + (T.fw[c] << SH));
                    if (T.rc[c] != T.fw[c]) v[e] += *reinterpret_cast<const uint16_t *>(colp + (T.rc[c] << SH));
                }
                uint4 o = make_uint4(v[0], v[1], v[2], v[3]);
                if (c0 != 0) {
                    const uint4 prev = out[c4];
                    o.x += prev.x; o.y += prev.y; o.z += prev.z; o.w += prev.w;
                }
                out[c4] = o;
            }
        }
        if (c1 >= ulast) break;
        lrb_barrier();
        clear();
    }
}

// ---------------------------------------------------------------------------
// k = 4 at STRIDE 2 (round 3).  A ds_add_u32 costs 4 cycles of the CU's LDS operand path whatever it adds to, and
// one per window is what held the kernel above at 94 % of that bound (33.8 % of the HBM roofline).  Every 4-mer is
// the prefix or the suffix of exactly one 5-mer that starts at an EVEN position, so the windows are tallied as
// 5-mers at positions 0, 2, 4, ... -- HALF the atomics -- into the 1,024-bin histogram of the k = 5 kernel (half
// groups: 32 reads, 64 KB, two workgroups to a CU), and the flush folds
//     n4[x] = sum_b h5[4 x + b]  (4-mers at even positions)  +  sum_a h5[256 a + x]  (4-mers at odd positions)
// into the canonical classes of count-kmers.cpp:38-64.  A read of even length ends with a 4-mer at an even
// position that no 5-mer holds: the lane that meets that position keeps its code in LDS (tail[column]) and the
// flush adds it.  5-mer starts are p <= L - 5; the 4-mer total is L - 3 as in count-kmers.cpp:80-86.
// ---------------------------------------------------------------------------
template <bool PRED, int KW = 5>   // KW: bases per window (the (k+1)-mer)
__device__ __forceinline__ void lane4s2_row(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t halo,
                                            uint32_t laneoff, uint32_t one, uint32_t pos0, uint32_t nk,
                                            uint32_t tailpos, uint32_t &tailv)
{
    constexpr int SH = 6; // a bin is 16 words (32 columns, two to a word)
    const uint32_t w[5] = {w0, w1, w2, w3, halo};
#pragma unroll
    for (int q = 0; q < 4; q += 2) {
        uint32_t t[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int qq = q + (i >> 3), p = 2 * (i & 7);
            const int used = 2 * p + 2 * KW;
            t[i] = used + SH <= 32 ? w[qq] >> (32 - used - SH) : __builtin_amdgcn_alignbit(w[qq], w[qq + 1], 64 - used - SH);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 16; ++i) t[i] = (t[i] & ((((1u << (2 * KW)) - 1u)) << SH)) | laneoff;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (!PRED) {
                lds_add(t[i], one);
            } else {
                const uint32_t pos = pos0 + (uint32_t)((q + (i >> 3)) * 16 + 2 * (i & 7));
                if (pos < nk) lds_add(t[i], one);
                else if (pos == tailpos) tailv = t[i];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// A workgroup walks half groups blockIdx.x, blockIdx.x + gridDim.x, ... : the first rows of the next half group and
// its lengths are requested before the current one is flushed, so that neither a dispatch nor the first memory
// latency stands between two groups (with one half group per workgroup a quarter of a workgroup's 15 us was that).
// STAMP (scripts/k4s2_probe.hip only): wave 0 of every workgroup writes s_memrealtime at the start of each half
// group's tally, at its end and after the flush into dbg[blockIdx.x * 256 ...].
// MANY (round 6): the groups of several resident batches behind ONE launch -- group_off holds a PAIR per group (first
// row, end row: a batch's last group does not end where the next batch's first one starts; rows count from the lowest
// of the batches' buffers), order[slot] the read's row in the merged output or ~0 for the padding of a batch's last
// group (lrb_packed_kmer_counts_many_dev builds both with one small kernel from a table of the batches).
template <int W, int NR, bool STAMP = false, int K = 4, bool MANY = false>
__global__ __launch_bounds__(64 * W) void k1_lane4s2_kernel(const uint4 *__restrict__ codes_t,
                                                            const uint64_t *__restrict__ group_off,
                                                            const uint32_t *__restrict__ order,
                                                            const uint32_t *__restrict__ lens, uint64_t n,
                                                            uint32_t *__restrict__ counts, uint64_t *dbg = nullptr)
{
    constexpr kmer_classes<K> T = make_kmer_classes<K>();
    constexpr int DIM = T.n;                  // 136 (k = 4) / 32 (k = 3)
    constexpr int U = 2, SH = 6;
    constexpr int KPOW = 1 << (2 * K);        // k-mers; the histogram's bins are the 4 KPOW (k+1)-mers
    constexpr int BINS = 4 * KPOW;
    constexpr int CLR = BINS / 16;            // 1-KiB slabs of the histogram
    // steps between flushes, a multiple of W * NR: a column takes 32 U tallies per step, and the flush adds every bin
    // TWICE (as a prefix and as a suffix) in packed 16-bit halves, so a column may hold 32,767 of them
    constexpr uint32_t CHUNK = (1020 / U / (W * NR)) * (W * NR);
    static_assert(DIM % 4 == 0 && CHUNK > 0 && (K == 3 || K == 4), "K");
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[]; // BINS x 16 words, then the flush's tables
    uint32_t *tail = smem + BINS * 16;                               // [32] closing k-mer of a column's read
    uint64_t *rcol = reinterpret_cast<uint64_t *>(tail + 32);        // [32] output row of a column's read
    uint32_t *cls = tail + 32 + 64;                                  // [DIM] fw | rc << 16
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t col = lane & 31u;                                 // the read's column in the histogram
    const uint32_t sub = lane >> 5;                                  // which row of the step's row pair
    if (threadIdx.x < (uint32_t)DIM) cls[threadIdx.x] = (uint32_t)T.fw[threadIdx.x] | ((uint32_t)T.rc[threadIdx.x] << 16);
    const uint64_t nhalf = ((n + 63) >> 6) << 1;
    const uint32_t laneoff = lds_addr_of(smem) + (col >> 1) * 4u;
    const uint32_t one = (col & 1u) ? 0x10000u : 1u;
    typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
    struct row_t {
        v4u_t w;
        uint32_t halo;
    };
    struct meta_t {     // what a half group's tally needs before its rows (no padding bytes: a struct with a trailing
        uint64_t r, row0;   // bool is copied through three bytes of scratch per lane)
        uint32_t L, rows, voff;
        uint32_t have;
    };
    auto load_meta = [&](uint64_t hg) -> meta_t {
        meta_t x;
        const uint64_t g = hg >> 1;
        const uint32_t in_group = (uint32_t)(hg & 1u) * 32u + col;
        const uint64_t slot = (g << 6) + in_group;
        x.have = slot < n;
        x.r = x.have ? (order ? order[slot] : slot) : 0;
        if (MANY && x.r == 0xFFFFFFFFull) x.have = 0, x.r = 0;   // (padding of a batch's last group)
        x.L = x.have ? lens[x.r] : 0u;
        x.row0 = group_off[MANY ? 2 * g : g];
        x.rows = (uint32_t)(group_off[MANY ? 2 * g + 1 : g + 1] - x.row0); // 1 + max rows
        x.voff = sub * 1024u + in_group * 16u;
        return x;
    };
    auto rsrc_of = [&](const meta_t &x, uint32_t j) {
        const char *base = reinterpret_cast<const char *>(codes_t) + x.row0 * 1024;
        const uint64_t left = j < x.rows ? (uint64_t)(x.rows - j) * 1024 : 0; // past the end: loads return 0
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base + (uint64_t)j * 1024), 0,
                                                 left < 0x7FFFFFFFull ? (int)left : 0x7FFFFFFF, 0x00020000);
    };
    auto load_row = [&](__amdgpu_buffer_rsrc_t rs, uint32_t voff, int imm) -> row_t {
        row_t x;
        x.w = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, imm, 0);
        x.halo = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, imm + 1024, 0);
        return x;
    };
    auto clear = [&]() {
        const uint4 z = {0u, 0u, 0u, 0u};
        uint4 *h4 = reinterpret_cast<uint4 *>(smem);
#pragma unroll
        for (int i = 0; i < CLR / W; ++i) h4[(i * W + wv) * 64 + lane] = z;
        if (CLR % W != 0 && (CLR / W) * W + wv < CLR) h4[((CLR / W) * W + wv) * 64 + lane] = z;
        if (threadIdx.x < 32) tail[threadIdx.x] = 0xFFFFFFFFu;
        lrb_barrier();
    };
    auto mine_below = [&](uint32_t bound) { return bound > wv ? (bound - wv + W - 1) / W : 0u; };

    uint64_t hg = blockIdx.x;
    if (hg >= nhalf) return;
    meta_t cur;
    row_t R[NR];
    {   // the rows first: they need the group's offset only, the read's length is two dependent loads away
        const uint64_t g = hg >> 1;
        cur.row0 = group_off[MANY ? 2 * g : g];
        cur.rows = (uint32_t)(group_off[MANY ? 2 * g + 1 : g + 1] - cur.row0);
        cur.voff = sub * 1024u + ((uint32_t)(hg & 1u) * 32u + col) * 16u;
        const auto rs = rsrc_of(cur, U * wv);
#pragma unroll
        for (int i = 0; i < NR; ++i) R[i] = load_row(rs, cur.voff, i * W * U * 1024);
        const meta_t full_meta = load_meta(hg);
        cur.have = full_meta.have;
        cur.r = full_meta.r;
        cur.L = full_meta.L;
    }
    clear();
    uint32_t n_stamp = 0;
    auto stamp = [&]() {
        if (STAMP && threadIdx.x == 0 && n_stamp < 255) dbg[(uint64_t)blockIdx.x * 256 + 1 + n_stamp++] = __builtin_amdgcn_s_memrealtime();
    };
    for (;;) {
        stamp();
        const uint64_t hg_next = hg + gridDim.x;
        const bool more = hg_next < nhalf;
        meta_t nxt = cur;
        if (more) nxt = load_meta(hg_next);     // asked for now, needed after this half group's tally
        if (threadIdx.x < 32) rcol[col] = cur.have ? cur.r : ~0ull;   // (the flush is several barriers away)
        const uint32_t L = cur.L;
        const uint32_t nk = L >= (uint32_t)(K + 1) ? L - (uint32_t)K : 0u;   // (k+1)-mer starts are the positions below nk
        // the read's last k-mer starts at L - k; when that position is even no (k+1)-mer holds it
        const uint32_t tailpos = (L >= (uint32_t)K && ((L - (uint32_t)K) & 1u) == 0u) ? L - (uint32_t)K : 0xFFFFFFFFu;
        const uint32_t last = cur.rows - 1;
        uint32_t nk_min = nk;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t other = __shfl_xor(nk_min, o, WAVE);
            nk_min = other < nk_min ? other : nk_min;
        }
        nk_min = __builtin_amdgcn_readfirstlane(nk_min);
        const uint32_t full = nk_min / 64 < last ? nk_min / 64 : last; // rows whose 64 positions are starts in every lane
        const uint32_t ufull = full / U;
        const uint32_t ulast = (last + U - 1) / U;
        const uint32_t mfull = mine_below(ufull);
        uint32_t m = 0;
        uint32_t tailv = 0xFFFFFFFFu, nouse = 0;
        for (uint32_t c0 = 0;; c0 += CHUNK) {
            const uint32_t c1 = c0 + CHUNK < ulast ? c0 + CHUNK : ulast; // this chunk: steps c0 .. c1
            const uint32_t mc1 = mine_below(c1);
            const uint32_t fast = mfull < mc1 ? mfull : mc1;
            for (; m + NR <= fast; m += NR) {
                const auto rs = rsrc_of(cur, U * (wv + W * (m + NR)));
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    lane4s2_row<false, K + 1>(R[i].w.x, R[i].w.y, R[i].w.z, R[i].w.w, R[i].halo, laneoff, one, 0u, 0u, 0u, nouse);
                    R[i] = load_row(rs, cur.voff, i * W * U * 1024);
                }
            }
            for (; m < mc1; ++m) {
                const uint32_t q = wv + W * m;
                if (q < ufull)
                    lane4s2_row<false, K + 1>(R[0].w.x, R[0].w.y, R[0].w.z, R[0].w.w, R[0].halo, laneoff, one, 0u, 0u, 0u, nouse);
                else
                    lane4s2_row<true, K + 1>(R[0].w.x, R[0].w.y, R[0].w.z, R[0].w.w, R[0].halo, laneoff, one,
                                      (U * q + sub) * 64u, nk, tailpos, tailv);
#pragma unroll
                for (int i = 0; i + 1 < NR; ++i) R[i] = R[i + 1];
                R[NR - 1] = load_row(rsrc_of(cur, U * (wv + W * (m + NR))), cur.voff, 0);
            }
            const bool final_chunk = c1 >= ulast;
            stamp();
            if (final_chunk && more) {   // the next half group's first rows travel while this one is flushed
                const auto rs = rsrc_of(nxt, U * wv);
#pragma unroll
                for (int i = 0; i < NR; ++i) R[i] = load_row(rs, nxt.voff, i * W * U * 1024);
            }
            // the closing 4-mer of an even-length read: the prefix of the 5-mer window its lane met at L - 4
            if (final_chunk && tailv != 0xFFFFFFFFu) tail[col] = (tailv >> (SH + 2)) & (uint32_t)(KPOW - 1); // bits SH.. hold the (k+1)-mer
            lrb_barrier();
            // (the flush and the clear that follows go ahead of the other workgroup's tally on this CU: the sooner
            // they are through, the sooner sixteen waves tally again)
            __builtin_amdgcn_s_setprio(2);
            lane4_flush_half<DIM, true, W, KPOW>(smem, tail, rcol, cls, counts, lane, wv, c0 != 0, final_chunk);
            stamp();
            if (final_chunk) break;
            lrb_barrier();
            clear();
            __builtin_amdgcn_s_setprio(0);
        }
        if (!more) break;
        lrb_barrier();
        clear();
        __builtin_amdgcn_s_setprio(0);
        cur = nxt;
        hg = hg_next;
    }
    if (STAMP && threadIdx.x == 0) {
        dbg[(uint64_t)blockIdx.x * 256] = n_stamp;
        stamp();
    }
}

// codes (per read) -> codes_t (group-transposed) through a 64-read x 16-row LDS tile: 256-B runs
// of one read in, 1-KiB rows of one 64-base column out.
__global__ __launch_bounds__(256) void codes_t_kernel(const uint32_t *__restrict__ codes,
                                                      const uint64_t *__restrict__ code_off,
                                                      const uint64_t *__restrict__ group_off,
                                                      const uint32_t *__restrict__ order, uint64_t n,
                                                      uint4 *__restrict__ codes_t)
{
    __shared__ uint4 tile[64][17];
    const uint64_t g = blockIdx.x;
    const uint64_t row0 = group_off[g];
    const uint32_t rows = (uint32_t)(group_off[g + 1] - row0);
    const uint32_t t = threadIdx.x;
    for (uint32_t j0 = 0; j0 < rows; j0 += 16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t rr = (t >> 4) + 16 * i, b = t & 15;
            const uint64_t slot = (g << 6) + rr;
            uint4 v = {0u, 0u, 0u, 0u};
            if (slot < n) {
                const uint64_t r = order ? order[slot] : slot;
                const uint64_t nw = code_off[r + 1] - code_off[r]; // words of this read's region (multiple of 4)
                if ((uint64_t)(j0 + b) * 4 < nw)
                    v = reinterpret_cast<const uint4 *>(codes + code_off[r])[j0 + b];
            }
            tile[rr][b] = v;
        }
        lrb_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t b = (t >> 6) + 4 * i, l = t & 63;
            if (j0 + b < rows) codes_t[(row0 + j0 + b) * 64 + l] = tile[l][b];
        }
        lrb_barrier();
    }
}

// codes (2 bits interleaved) -> bit planes {H, L} per 32-base block, stored as uint2 at
// word 2*(mask_off[r] + block).  One lane per block.
__device__ __forceinline__ uint32_t odd_bits16(uint32_t w)
{
    uint32_t x = (w >> 1) & 0x55555555u;
    x = (x | (x >> 1)) & 0x33333333u;
    x = (x | (x >> 2)) & 0x0F0F0F0Fu;
    x = (x | (x >> 4)) & 0x00FF00FFu;
    x = (x | (x >> 8)) & 0x0000FFFFu;
    return x;
}

__global__ __launch_bounds__(256) void planes_kernel(const uint32_t *__restrict__ codes,
                                                     const uint64_t *__restrict__ code_off,
                                                     const uint64_t *__restrict__ mask_off,
                                                     uint64_t n, uint32_t *__restrict__ planes)
{
    const uint32_t lane = lane_id();
    const uint64_t wave0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = wave0; r < n; r += nwaves) {
        const uint32_t *cw = codes + code_off[r];
        const uint64_t ncw = code_off[r + 1] - code_off[r];
        const uint64_t nmw = mask_off[r + 1] - mask_off[r];
        uint2 *out = reinterpret_cast<uint2 *>(planes + 2 * mask_off[r]);
        for (uint64_t b = lane; b < nmw; b += WAVE) {
            uint32_t c0 = 0, c1 = 0;
            if (2 * b + 1 < ncw) {
                c0 = cw[2 * b];
                c1 = cw[2 * b + 1];
            }
            uint2 v;
            v.x = (odd_bits16(c0) << 16) | odd_bits16(c1);
            v.y = (odd_bits16(c0 << 1) << 16) | odd_bits16(c1 << 1);
            out[b] = v;
        }
    }
}

// ---------------------------------------------------------------------------
// 15-mer window helpers shared by K2 and K3.  One lane owns a 32-base chunk.
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k15_accum_kernel(const uint32_t *__restrict__ codes,
                                                        const uint32_t *__restrict__ mask,
                                                        const uint64_t *__restrict__ code_off,
                                                        const uint64_t *__restrict__ mask_off,
                                                        const uint32_t *__restrict__ lens,
                                                        uint64_t n, uint32_t *__restrict__ table)
{
    const uint32_t lane = lane_id();
    const uint64_t wave0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = wave0; r < n; r += nwaves) {
        const uint32_t L = lens[r];
        if (L < 15) continue;
        const uint32_t *cw = codes + code_off[r];
        const uint32_t *mw = mask + mask_off[r];
        const uint32_t nchunks = (L + 31) >> 5;
        for (uint32_t c = lane; c < nchunks; c += WAVE) {
            const uint32_t vm = valid15_starts(mw[c], mw[c + 1]);
            if (!vm) continue;
            const uint32_t w0 = cw[2 * c], w1 = cw[2 * c + 1], w2 = cw[2 * c + 2];
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                if (vm & (0x80000000u >> i)) {
                    const uint32_t val = i < 16 ? k15_at(w0, w1, i) : k15_at(w1, w2, i - 16);
                    atomicAdd(&table[val], 1u);
                }
            }
        }
    }
}

// rc of n 2-bit groups: reverse the groups, complement each (XOR 10b)
__device__ __forceinline__ uint32_t rc_groups(uint32_t v, int ngroups)
{
    uint32_t r = __builtin_bitreverse32(v);                        // bit reversal
    r = ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);       // un-swap inside groups
    r >>= (32 - 2 * ngroups);
    return r ^ (0xAAAAAAAAu >> (32 - 2 * ngroups));
}

// T[x] = F[x] + F[rc(x)] in place.  x = [t:6][m:18][l:6]; rc(x) = [rc3(l)][rc9(m)][rc3(t)].
// A workgroup owns the tile pair (m, rc9(m)), m < rc9(m) (9 groups: never equal): two
// 64x64 tiles whose rows are 256-B contiguous in HBM on both sides of the transpose.
__global__ __launch_bounds__(256) void k15_mirror_kernel(uint32_t *__restrict__ table)
{
    __shared__ uint32_t A[64][65];
    __shared__ uint32_t B[64][65];
    const uint32_t m = blockIdx.x;
    const uint32_t mr = rc_groups(m, 9);
    if (m > mr) return;
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    uint32_t *pa = table + ((uint64_t)m << 6);
    uint32_t *pb = table + ((uint64_t)mr << 6);
    for (uint32_t t = wave; t < 64; t += 4) {
        A[t][lane] = pa[((uint64_t)t << 24) + lane];
        B[t][lane] = pb[((uint64_t)t << 24) + lane];
    }
    lrb_barrier();
    const uint32_t lr = rc_groups(lane, 3);
    for (uint32_t t = wave; t < 64; t += 4) {
        const uint32_t tr = rc_groups(t, 3);
        pa[((uint64_t)t << 24) + lane] = A[t][lane] + B[lr][tr];
        pb[((uint64_t)t << 24) + lane] = B[t][lane] + A[lr][tr];
    }
}

// The multi-GPU form of the same step: T = sum over ranks of (F_r + F_r o rc) is
// determined by its CANONICAL HALF.  x and rc(x) differ in the high code bit of the middle base
// (k = 15 is odd; complement = XOR 10b), which is bit 15 of x = bit 9 of m: the canonical member of
// each pair is the one with that bit 0, and dropping the bit numbers the 2^29 pairs densely,
//     h = [t:6][m':17][l:6],  m' = m without its bit 9.
// fold:   H[h] = F[x] + F[rc(x)]   (4 GiB read, 2 GiB written)  -- all-reduce 2 GiB instead of 4 --
// expand: T[x] = T[rc(x)] = H[h]   (2 GiB read, 4 GiB written).
// A workgroup owns one pair of 64 x 64 tiles as in the mirror kernel.
__device__ __forceinline__ uint32_t k15_canon_m(uint32_t mp) { return ((mp >> 9) << 10) | (mp & 0x1FFu); }

__global__ __launch_bounds__(256) void k15_fold_half_kernel(const uint32_t *__restrict__ table,
                                                            uint32_t *__restrict__ half)
{
    __shared__ uint32_t A[64][65];
    __shared__ uint32_t B[64][65];
    const uint32_t mp = blockIdx.x;
    const uint32_t m = k15_canon_m(mp), mr = rc_groups(m, 9);
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    const uint32_t *pa = table + ((uint64_t)m << 6);
    const uint32_t *pb = table + ((uint64_t)mr << 6);
    for (uint32_t t = wave; t < 64; t += 4) {
        A[t][lane] = pa[((uint64_t)t << 24) + lane];
        B[t][lane] = pb[((uint64_t)t << 24) + lane];
    }
    lrb_barrier();
    const uint32_t lr = rc_groups(lane, 3);
    uint32_t *ph = half + ((uint64_t)mp << 6);
    for (uint32_t t = wave; t < 64; t += 4) {
        const uint32_t tr = rc_groups(t, 3);
        ph[((uint64_t)t << 23) + lane] = A[t][lane] + B[lr][tr];
    }
}

__global__ __launch_bounds__(256) void k15_expand_half_kernel(const uint32_t *__restrict__ half,
                                                              uint32_t *__restrict__ table)
{
    __shared__ uint32_t A[64][65];
    const uint32_t mp = blockIdx.x;
    const uint32_t m = k15_canon_m(mp), mr = rc_groups(m, 9);
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    const uint32_t *ph = half + ((uint64_t)mp << 6);
    for (uint32_t t = wave; t < 64; t += 4) A[t][lane] = ph[((uint64_t)t << 23) + lane];
    lrb_barrier();
    const uint32_t lr = rc_groups(lane, 3);
    uint32_t *pa = table + ((uint64_t)m << 6);
    uint32_t *pb = table + ((uint64_t)mr << 6);
    for (uint32_t t = wave; t < 64; t += 4) {
        const uint32_t tr = rc_groups(t, 3);
        pa[((uint64_t)t << 24) + lane] = A[t][lane];
        pb[((uint64_t)t << 24) + lane] = A[lr][tr];
    }
}

// ---------------------------------------------------------------------------
// K3: coverage histogram.  One wave per read; per wave an LDS histogram
// [bin][sub] like K1.  Gathers are issued for a whole 32-base chunk before any
// is consumed (32 independent loads in flight per lane).
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(256) void cov_hist_kernel(const uint32_t *__restrict__ codes,
                                                       const uint32_t *__restrict__ mask,
                                                       const uint64_t *__restrict__ code_off,
                                                       const uint64_t *__restrict__ mask_off,
                                                       const uint32_t *__restrict__ lens,
                                                       uint64_t n,
                                                       const uint32_t *__restrict__ table,
                                                       uint32_t bs, uint32_t bins, uint32_t sub_log2,
                                                       uint32_t *__restrict__ hist_out,
                                                       uint32_t *__restrict__ sums_out)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    const uint32_t hwords = bins << sub_log2;
    uint32_t *h = smem + wave * hwords;
    const uint32_t subs = 1u << sub_log2;
    const uint32_t sub = lane & (subs - 1);
    for (uint64_t r = (uint64_t)blockIdx.x * 4 + wave; r < n; r += (uint64_t)gridDim.x * 4) {
        for (uint32_t i = lane; i < hwords; i += WAVE) h[i] = 0;
        wave_lds_fence();
        const uint32_t L = lens[r];
        uint32_t nvalid = 0;
        if (L >= 15) {
            const uint32_t *cw = codes + code_off[r];
            const uint32_t *mw = mask + mask_off[r];
            const uint32_t nchunks = (L + 31) >> 5;
            for (uint32_t c = lane; c < nchunks; c += WAVE) {
                const uint32_t vm = valid15_starts(mw[c], mw[c + 1]);
                if (!vm) continue;
                nvalid += __popc(vm);
                const uint32_t w0 = cw[2 * c], w1 = cw[2 * c + 1], w2 = cw[2 * c + 2];
                uint32_t cnt[32];
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const uint32_t val = i < 16 ? k15_at(w0, w1, i) : k15_at(w1, w2, i - 16);
                    cnt[i] = table[val]; // always in range; tally is predicated below
                }
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    if (vm & (0x80000000u >> i)) {
                        const uint32_t b = cov_bin_dev(cnt[i], bs, bins);
                        atomicAdd(&h[(b << sub_log2) + sub], 1u);
                    }
                }
            }
        }
        wave_lds_fence();
        for (uint32_t b = lane; b < bins; b += WAVE) {
            uint32_t s = 0;
            for (uint32_t q = 0; q < subs; ++q) s += h[(b << sub_log2) + ((q + lane) & (subs - 1))];
            hist_out[r * bins + b] = s;
        }
        nvalid = wave_sum_u32(nvalid);
        if (lane == 0) sums_out[r] = nvalid;
        wave_lds_fence();
    }
}

// K3 on a COMPACT map of the table.  The gathers of cov_hist_kernel are random 4-byte reads of a
// 4 GiB table: every one is an HBM sector (rocprofv3: 64 B fetched per gather, 0.1 % L2 hits,
// profiles/r01_k2_k3_rocprof_summary.txt).  What line_to_vec needs of a count is only its BIN
// (kmer_utils.h:55-69), and the mirrored table is symmetric, T[x] == T[rc(x)].  So one streaming
// pass turns the table into bin ids of the canonical half, one BYTE per pair (x, rc(x)):
//     map[h] = cov_bin(T[x]),  x the member with bit 15 clear, h = x without that bit   (2^29 B = 512 MB)
// -- an eighth of the footprint, half of which the 256 MB Infinity Cache holds.  The query side pays
// a reverse complement per window on vector ALUs that were idle anyway.  Same histograms bit for bit.
__global__ __launch_bounds__(256) void cov_map_build_kernel(const uint32_t *__restrict__ table, uint32_t bs,
                                                            uint32_t bins, uint8_t *__restrict__ map)
{
    const uint64_t nvec = LRB_K15_HALF_ENTRIES / 16;
    for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec;
         v += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t h = v << 4;
        const uint64_t x = ((h >> 15) << 16) | (h & 0x7FFFu);
        const uint4 *src = reinterpret_cast<const uint4 *>(table + x);
        uint32_t o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 t = src[q];
            o[q] = cov_bin_dev(t.x, bs, bins) | (cov_bin_dev(t.y, bs, bins) << 8) | (cov_bin_dev(t.z, bs, bins) << 16) |
                   (cov_bin_dev(t.w, bs, bins) << 24);
        }
        reinterpret_cast<uint4 *>(map)[v] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

__device__ __forceinline__ uint32_t cov_map_index(uint32_t val)
{
    const uint32_t r = rc_groups(val, 15);
    const uint32_t x = (val & 0x8000u) ? r : val; // the strand whose middle base has high code bit 0
    return ((x >> 16) << 15) | (x & 0x7FFFu);
}

// the same from the reverse complement of a whole SPAN: with R = (rc32(lo) : rc32(hi)) the reverse complement of the
// 15-mer starting at base q of (hi : lo) is (R >> 2q) & mask -- one v_alignbit per window instead of a bit reversal

__global__ __launch_bounds__(256) void cov_hist_map_kernel(const uint32_t *__restrict__ codes,
                                                           const uint32_t *__restrict__ mask,
                                                           const uint64_t *__restrict__ code_off,
                                                           const uint64_t *__restrict__ mask_off,
                                                           const uint32_t *__restrict__ lens, uint64_t n,
                                                           const uint8_t *__restrict__ map, uint32_t bins,
                                                           uint32_t sub_log2, uint32_t min_len,
                                                           uint32_t *__restrict__ hist_out,
                                                           uint32_t *__restrict__ sums_out)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    const uint32_t hwords = bins << sub_log2;
    uint32_t *h = smem + wave * hwords;
    const uint32_t subs = 1u << sub_log2;
    const uint32_t sub = lane & (subs - 1);
    const __amdgpu_buffer_rsrc_t map_rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(map), 0, (int)LRB_COV_MAP_BYTES, 0x00020000);
    for (uint64_t r = (uint64_t)blockIdx.x * 4 + wave; r < n; r += (uint64_t)gridDim.x * 4) {
        const uint32_t L = lens[r];
        if (L < min_len) continue; // the sweep form leaves only its over-long reads to this kernel
        for (uint32_t i = lane; i < hwords; i += WAVE) h[i] = 0;
        wave_lds_fence();
        uint32_t nvalid = 0;
        if (L >= 15) {
            const uint32_t *cw = codes + code_off[r];
            const uint32_t *mw = mask + mask_off[r];
            const uint32_t nchunks = (L + 31) >> 5;
            for (uint32_t c = lane; c < nchunks; c += WAVE) {
                const uint32_t vm = valid15_starts(mw[c], mw[c + 1]);
                if (!vm) continue;
                nvalid += __popc(vm);
                const uint32_t w0 = cw[2 * c], w1 = cw[2 * c + 1], w2 = cw[2 * c + 2];
                uint32_t b8[32];
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const uint32_t val = i < 16 ? k15_at(w0, w1, i) : k15_at(w1, w2, i - 16);
                    // always in range; the tally is predicated below.  (The cache-policy bits of the load --
                    // sc0, nt, sc1 in any combination -- change nothing: every gather is one 128-byte line
                    // fill either way, profiles/r02_k3_rocprof_summary.txt)
                    b8[i] = __builtin_amdgcn_raw_buffer_load_b8(map_rs, (int)cov_map_index(val), 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 32; ++i)
                    if (vm & (0x80000000u >> i)) atomicAdd(&h[(b8[i] << sub_log2) + sub], 1u);
            }
        }
        wave_lds_fence();
        for (uint32_t b = lane; b < bins; b += WAVE) {
            uint32_t s = 0;
            for (uint32_t q = 0; q < subs; ++q) s += h[(b << sub_log2) + ((q + lane) & (subs - 1))];
            hist_out[r * bins + b] = s;
        }
        nvalid = wave_sum_u32(nvalid);
        if (lane == 0) sums_out[r] = nvalid;
        wave_lds_fence();
    }
}

// (K2 and K3 on one partition of the windows -- part / order / tally / sweep -- live in lrb_lists.hip)
#define CJ_MAX_WINDOWS 65535u // reads of more windows are not in the window lists (u16 histogram counters)

// H[h] += 1 for every valid 15-mer of reads of at least min_len bases, one atomic each: the reads the slice lists
// leave out (more than 65,535 windows), and small batches altogether (min_len = 0)
__global__ __launch_bounds__(256) void k15_accum_half_kernel(const uint32_t *__restrict__ codes,
                                                             const uint32_t *__restrict__ mask,
                                                             const uint64_t *__restrict__ code_off,
                                                             const uint64_t *__restrict__ mask_off,
                                                             const uint32_t *__restrict__ lens, uint64_t n,
                                                             uint32_t min_len, uint32_t *__restrict__ half)
{
    const uint32_t lane = lane_id();
    const uint64_t wave0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = wave0; r < n; r += nwaves) {
        const uint32_t L = lens[r];
        if (L < 15 || L < min_len) continue;
        const uint32_t *cw = codes + code_off[r];
        const uint32_t *mw = mask + mask_off[r];
        const uint32_t nchunks = (L + 31) >> 5;
        for (uint32_t c = lane; c < nchunks; c += WAVE) {
            const uint32_t vm = valid15_starts(mw[c], mw[c + 1]);
            if (!vm) continue;
            const uint32_t c0 = cw[2 * c], c1 = cw[2 * c + 1], c2 = cw[2 * c + 2];
            const uint32_t q0 = rc32(c0), q1 = rc32(c1), q2 = rc32(c2);
#pragma unroll
            for (int i = 0; i < 32; ++i)
                if (vm & (0x80000000u >> i)) {
                    const uint32_t val = i < 16 ? k15_at(c0, c1, i) : k15_at(c1, c2, i - 16);
                    const uint32_t rc = (i < 16 ? __builtin_amdgcn_alignbit(q1, q0, 2 * i)
                                                : __builtin_amdgcn_alignbit(q2, q1, 2 * (i - 16))) & K15_MASK;
                    atomicAdd(&half[cov_map_index_rc(val, rc)], 1u);
                }
        }
    }
}

// the compact map straight from the canonical half: map[h] = cov_bin(H[h])  (cov_map_build_kernel reads the same
// numbers out of the mirrored table)
__global__ __launch_bounds__(256) void cov_map_build_half_kernel(const uint32_t *__restrict__ half, uint32_t bs,
                                                                 uint32_t bins, uint8_t *__restrict__ map)
{
    const uint64_t nvec = LRB_K15_HALF_ENTRIES / 16;
    for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec;
         v += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 *src = reinterpret_cast<const uint4 *>(half + (v << 4));
        uint32_t o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 t = src[q];
            o[q] = cov_bin_dev(t.x, bs, bins) | (cov_bin_dev(t.y, bs, bins) << 8) | (cov_bin_dev(t.z, bs, bins) << 16) |
                   (cov_bin_dev(t.w, bs, bins) << 24);
        }
        reinterpret_cast<uint4 *>(map)[v] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// ===========================================================================
// C ABI (device half)
// ===========================================================================
static thread_local char g_err[512] = "";

void lrb_set_error(const char *fmt, const char *a, const char *b)
{
    snprintf(g_err, sizeof g_err, fmt, a ? a : "", b ? b : "");
}

extern "C" const char *lrb_last_error(void) { return g_err; }
extern "C" int lrb_version(void) { return 100; }

static int ws_get(lrb_ctx *c, int slot, uint64_t bytes, void **p) { return lrb_ws_get(c, slot, bytes, p); }

extern "C" int lrb_device_count(int *count)
{
    ARG_TRY(count != nullptr);
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *count = 0;
        lrb_set_error("hipGetDeviceCount failed: %s%s", hipGetErrorString(e), "");
        return LRB_ERR_NODEVICE;
    }
    *count = c;
    return LRB_OK;
}

static uint64_t rc_host(uint64_t x, unsigned k)
{
    uint64_t r = 0;
    for (unsigned i = 0; i < k; i++) r = (r << 2) | (((x >> (2 * i)) & 3u) ^ 2u);
    return r;
}

// Canonical numbering: ascending codes, a code takes its reverse complement's
// number when that one came first (count-kmers.cpp:38-64).
extern "C" int lrb_kmer_lut(int k, uint32_t *lut, uint32_t *dim)
{
    ARG_TRY(k >= 3 && k <= 5);
    ARG_TRY(lut != nullptr);
    const uint32_t ncodes = 1u << (2 * k);
    uint32_t next = 0;
    for (uint32_t c = 0; c < ncodes; ++c) {
        const uint32_t rc = (uint32_t)rc_host(c, k);
        lut[c] = rc < c ? lut[rc] : next++;
    }
    if (dim) *dim = next;
    return LRB_OK;
}

extern "C" int lrb_kmer_dim(int k, uint32_t *dim)
{
    ARG_TRY(k >= 3 && k <= 5);
    ARG_TRY(dim != nullptr);
    uint32_t lut[1024];
    return lrb_kmer_lut(k, lut, dim);
}

extern "C" int lrb_ctx_create(int device, void *stream, int own_stream, lrb_ctx **out)
{
    ARG_TRY(out != nullptr);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        lrb_set_error("no HIP device visible%s%s", "", "");
        return LRB_ERR_NODEVICE;
    }
    ARG_TRY(device >= 0 && device < count);
    HIP_TRY(hipSetDevice(device));
    lrb_ctx *c = (lrb_ctx *)calloc(1, sizeof(lrb_ctx));
    if (!c) return LRB_ERR_NOMEM;
    c->device = device;
    if (!own_stream) {
        c->stream = (hipStream_t)stream;
        c->own_stream = false;
    } else {
        HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    for (int k = 3; k <= 5; ++k) {
        uint32_t lut[1024];
        uint16_t lut16[1024];
        lrb_kmer_lut(k, lut, &c->dim[k]);
        for (int i = 0; i < (1 << (2 * k)); ++i) lut16[i] = (uint16_t)lut[i];
        HIP_TRY(hipMalloc((void **)&c->d_lut[k], sizeof(uint16_t) << (2 * k)));
        HIP_TRY(hipMemcpy(c->d_lut[k], lut16, sizeof(uint16_t) << (2 * k), hipMemcpyHostToDevice));
    }
    *out = c;
    return LRB_OK;
}

extern "C" int lrb_ctx_destroy(lrb_ctx *c)
{
    if (!c) return LRB_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (int k = 3; k <= 5; ++k)
        if (c->d_lut[k]) (void)hipFree(c->d_lut[k]);
    for (int i = 0; i < LRB_WS_SLOTS; ++i)
        if (c->ws[i]) (void)hipFree(c->ws[i]);
    for (int i = 0; i < LRB_POOL_SLOTS; ++i)
        if (c->pool_ptr[i]) (void)hipFree(c->pool_ptr[i]);
    lrb_resident_lists_drop(c);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->stage_ev_live) (void)hipEventDestroy(c->stage_ev);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    free(c);
    return LRB_OK;
}

// ---- the list pool (see lrb_device.h) ----
static void pool_drop(lrb_ctx *c, uint64_t at_least)
{
    for (int i = 0; i < LRB_POOL_SLOTS; ++i)
        if (c->pool_ptr[i] && c->pool_size[i] >= at_least) {
            (void)hipFree(c->pool_ptr[i]);
            c->pool_held -= c->pool_size[i];
            c->pool_ptr[i] = nullptr;
            c->pool_size[i] = 0;
        }
}

// `bytes` of device memory: a retained block of that size or up to a quarter more, else a new allocation; *got = its size
static hipError_t pool_take(lrb_ctx *c, uint64_t bytes, void **p, uint64_t *got)
{
    int best = -1;
    for (int i = 0; i < LRB_POOL_SLOTS; ++i)
        if (c->pool_ptr[i] && c->pool_size[i] >= bytes && c->pool_size[i] <= bytes + bytes / 4 &&
            (best < 0 || c->pool_size[i] < c->pool_size[best]))
            best = i;
    if (best >= 0) {
        *p = c->pool_ptr[best];
        *got = c->pool_size[best];
        c->pool_held -= c->pool_size[best];
        c->pool_ptr[best] = nullptr;
        c->pool_size[best] = 0;
        return hipSuccess;
    }
    *got = bytes;
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess && c->pool_held) { // what is retained may be what is missing
        (void)hipGetLastError();
        pool_drop(c, 0);
        e = hipMalloc(p, bytes);
    }
    return e;
}

static void pool_give(lrb_ctx *c, void *p, uint64_t bytes)
{
    if (!p) return;
    if (c && bytes >= (1ull << 20) && c->pool_held + bytes <= c->pool_cap)
        for (int i = 0; i < LRB_POOL_SLOTS; ++i)
            if (!c->pool_ptr[i]) {
                c->pool_ptr[i] = p;
                c->pool_size[i] = bytes;
                c->pool_held += bytes;
                return;
            }
    (void)hipFree(p);
}

extern "C" int lrb_ctx_list_pool(lrb_ctx *c, uint64_t max_bytes)
{
    ARG_TRY(c != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    c->pool_cap = max_bytes;
    if (c->pool_held > max_bytes) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        pool_drop(c, 0);
    }
    return LRB_OK;
}

extern "C" int lrb_ctx_sync(lrb_ctx *c)
{
    ARG_TRY(c != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return LRB_OK;
}

extern "C" int lrb_ctx_trim(lrb_ctx *c, uint64_t keep_below)
{
    ARG_TRY(c != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    lrb_resident_lists_drop(c);
    ++c->lists_epoch;
    for (int i = 0; i < LRB_WS_SLOTS; ++i)
        if (c->ws[i] && c->ws_bytes[i] >= keep_below) {
            HIP_TRY(hipFree(c->ws[i]));
            c->ws[i] = nullptr;
            c->ws_bytes[i] = 0;
        }
    pool_drop(c, keep_below);
    return LRB_OK;
}

// (diagnostics) where workspace slot `slot` lives right now and how large it is; valid until the next call that uses it
extern "C" int lrb_ctx_ws_info(const lrb_ctx *c, int slot, void **d_ptr, uint64_t *bytes)
{
    ARG_TRY(c != nullptr && slot >= 0 && slot < LRB_WS_SLOTS && d_ptr != nullptr && bytes != nullptr);
    *d_ptr = c->ws[slot];
    *bytes = c->ws_bytes[slot];
    return LRB_OK;
}

extern "C" int lrb_ctx_partition_retries(const lrb_ctx *c, uint64_t *count)
{
    ARG_TRY(c != nullptr && count != nullptr);
    *count = c->wl_retries;
    return LRB_OK;
}

extern "C" int lrb_ctx_stream(lrb_ctx *c, void **stream)
{
    ARG_TRY(c != nullptr && stream != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    *stream = (void *)c->stream;
    return LRB_OK;
}

extern "C" int lrb_dev_alloc(lrb_ctx *c, uint64_t bytes, void **d_ptr)
{
    ARG_TRY(c != nullptr && d_ptr != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMalloc(d_ptr, bytes ? bytes : 16));
    return LRB_OK;
}

extern "C" int lrb_dev_free(lrb_ctx *c, void *d_ptr)
{
    ARG_TRY(c != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    if (d_ptr) HIP_TRY(hipFree(d_ptr));
    return LRB_OK;
}

extern "C" int lrb_dev_mem_info(lrb_ctx *c, uint64_t *free_bytes, uint64_t *total_bytes)
{
    ARG_TRY(c != nullptr && free_bytes != nullptr && total_bytes != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    size_t f = 0, t = 0;
    HIP_TRY(hipMemGetInfo(&f, &t));
    *free_bytes = f;
    *total_bytes = t;
    return LRB_OK;
}

extern "C" int lrb_host_alloc(lrb_ctx *c, uint64_t bytes, void **h_ptr)
{
    ARG_TRY(c != nullptr && h_ptr != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipHostMalloc(h_ptr, bytes ? bytes : 1, hipHostMallocDefault));
    return LRB_OK;
}

extern "C" int lrb_host_free(lrb_ctx *c, void *h_ptr)
{
    ARG_TRY(c != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    if (h_ptr) HIP_TRY(hipHostFree(h_ptr));
    return LRB_OK;
}

extern "C" int lrb_dev_memset(lrb_ctx *c, void *d_ptr, int value, uint64_t bytes)
{
    ARG_TRY(c != nullptr && (d_ptr != nullptr || bytes == 0));
    HIP_TRY(hipSetDevice(c->device));
    if (bytes) HIP_TRY(hipMemsetAsync(d_ptr, value, bytes, c->stream));
    return LRB_OK;
}

extern "C" int lrb_copy_h2d(lrb_ctx *c, void *d_dst, const void *src, uint64_t bytes)
{
    ARG_TRY(c != nullptr && (bytes == 0 || (d_dst && src)));
    HIP_TRY(hipSetDevice(c->device));
    if (bytes) {
        HIP_TRY(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return LRB_OK;
}

int lrb_stage_upload(lrb_ctx *c, int slot, const void *src, uint64_t bytes, void **d_ptr)
{
    int rc = lrb_ws_get(c, slot, bytes + 64, d_ptr);
    if (rc != LRB_OK) return rc;
    if (c->stage_ev_live) HIP_TRY(hipEventSynchronize(c->stage_ev));   // the last upload has left the staging
    if (c->h_stage_bytes < bytes) {
        if (c->h_stage) (void)hipHostFree(c->h_stage);
        c->h_stage = nullptr;
        c->h_stage_bytes = 0;
        const uint64_t want = bytes + (bytes >> 1) + 4096;
        HIP_TRY(hipHostMalloc(&c->h_stage, want, hipHostMallocDefault));
        c->h_stage_bytes = want;
    }
    if (!c->stage_ev_live) {
        HIP_TRY(hipEventCreateWithFlags(&c->stage_ev, hipEventDisableTiming));
        c->stage_ev_live = true;
    }
    memcpy(c->h_stage, src, bytes);
    HIP_TRY(hipMemcpyAsync(*d_ptr, c->h_stage, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipEventRecord(c->stage_ev, c->stream));
    return LRB_OK;
}

extern "C" int lrb_copy_d2h(lrb_ctx *c, void *dst, const void *d_src, uint64_t bytes)
{
    ARG_TRY(c != nullptr && (bytes == 0 || (dst && d_src)));
    HIP_TRY(hipSetDevice(c->device));
    if (bytes) {
        HIP_TRY(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return LRB_OK;
}

// ---- layout ---------------------------------------------------------------
static inline uint64_t code_region_words(uint64_t L) { return ((((L + 15) >> 4) + 3) & ~3ull) + 4; }
static inline uint64_t mask_region_words(uint64_t L) { return ((((L + 31) >> 5) + 3) & ~3ull) + 4; }

extern "C" int lrb_pack_layout(const uint64_t *offs, uint64_t n, uint32_t *lens,
                               uint64_t *code_off, uint64_t *mask_off)
{
    ARG_TRY(offs != nullptr && code_off != nullptr && mask_off != nullptr);
    uint64_t co = 0, mo = 0;
    for (uint64_t r = 0; r < n; ++r) {
        ARG_TRY(offs[r + 1] >= offs[r]);
        const uint64_t L = offs[r + 1] - offs[r];
        if (L >= 0xFFFFFFFFull) {
            lrb_set_error("read too long for the packed layout (>= 2^32-1 bases)%s%s", "", "");
            return LRB_ERR_ARG;
        }
        if (lens) lens[r] = (uint32_t)L;
        code_off[r] = co;
        mask_off[r] = mo;
        co += code_region_words(L);
        mo += mask_region_words(L);
    }
    code_off[n] = co;
    mask_off[n] = mo;
    return LRB_OK;
}

static int grid_for_waves(const lrb_ctx *c, uint64_t n_waves_wanted, int waves_per_block,
                          int blocks_per_cu)
{
    uint64_t blocks = (n_waves_wanted + waves_per_block - 1) / waves_per_block;
    const uint64_t cap = (uint64_t)c->n_cu * blocks_per_cu;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

extern "C" int lrb_pack_reads_dev(lrb_ctx *c, const uint8_t *d_seqs, uint64_t seq_bytes,
                                  const uint64_t *d_offs, uint64_t n,
                                  const uint64_t *d_code_off, const uint64_t *d_mask_off,
                                  uint32_t *d_codes, uint32_t *d_mask, uint32_t *d_planes)
{
    ARG_TRY(c != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    (void)seq_bytes;
    if (n == 0) return LRB_OK;
    ARG_TRY(d_seqs && d_offs && d_code_off && d_codes);
    ARG_TRY((d_mask == nullptr && d_planes == nullptr) || d_mask_off != nullptr);
    const int grid = grid_for_waves(c, n, 4, 8);
    hipLaunchKernelGGL(pack_kernel, dim3(grid), dim3(256), 0, c->stream, d_seqs, d_offs, n,
                       d_code_off, d_mask_off, d_codes, d_mask, d_planes);
    HIP_TRY(hipGetLastError());
    return LRB_OK;
}

// ---- K1 --------------------------------------------------------------------
template <int K, int SUBS, int WPS, int TW = 2>
static int launch_k1(lrb_ctx *c, const uint32_t *d_codes, const uint64_t *d_code_off,
                     const uint32_t *d_lens, uint64_t n, uint32_t *d_counts)
{
    constexpr int BINS = 1 << (2 * K);
    const uint32_t dimpad = (c->dim[K] + 63u) & ~63u;
    const size_t smem = (size_t)4 * BINS * SUBS * 4 + (size_t)4 * dimpad * 4 + BINS * 2;
    static lrb_per_device_once attr_done;
    if (attr_done.need(c->device)) {
        HIP_TRY(hipFuncSetAttribute((const void *)k1_count_kernel<K, SUBS, WPS, TW>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    }
    int per_cu = (int)((160 * 1024) / smem);
    if (per_cu > 2 * WPS) per_cu = 2 * WPS; // 4 waves per block, WPS waves per SIMD
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    const int grid = grid_for_waves(c, n, 4, per_cu);
    hipLaunchKernelGGL((k1_count_kernel<K, SUBS, WPS, TW>), dim3(grid), dim3(256), smem, c->stream,
                       d_codes, d_code_off, d_lens, n, c->d_lut[K], c->dim[K], dimpad, d_counts);
    HIP_TRY(hipGetLastError());
    return LRB_OK;
}

extern "C" int lrb_kmer_counts_dev(lrb_ctx *c, const uint32_t *d_codes, const uint64_t *d_code_off,
                                   const uint32_t *d_lens, uint64_t n, int k, uint32_t *d_counts)
{
    ARG_TRY(c != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    ARG_TRY(k >= 3 && k <= 5);
    if (n == 0) return LRB_OK;
    ARG_TRY(d_codes && d_code_off && d_lens && d_counts);
    // sub-counters per bin x waves per SIMD, measured: more sub-counters (fewer conflicts) at half the occupancy lose
    // 18-43 %; four word pairs per lane per trip instead of two lose 7-8 % (tail of the read)
    switch (k) {
    case 3: return launch_k1<3, 16, 8>(c, d_codes, d_code_off, d_lens, n, d_counts);
    case 4: return launch_k1<4, 8, 4>(c, d_codes, d_code_off, d_lens, n, d_counts);
    default: return launch_k1<5, 4, 2>(c, d_codes, d_code_off, d_lens, n, d_counts);
    }
}

extern "C" int lrb_planes_from_codes_dev(lrb_ctx *c, const uint32_t *d_codes,
                                         const uint64_t *d_code_off, const uint64_t *d_mask_off,
                                         uint64_t n, uint32_t *d_planes)
{
    ARG_TRY(c != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    if (n == 0) return LRB_OK;
    ARG_TRY(d_codes && d_code_off && d_mask_off && d_planes);
    const int grid = grid_for_waves(c, n, 4, 8);
    hipLaunchKernelGGL(planes_kernel, dim3(grid), dim3(256), 0, c->stream, d_codes, d_code_off,
                       d_mask_off, n, d_planes);
    HIP_TRY(hipGetLastError());
    return LRB_OK;
}

extern "C" int lrb_kmer_counts3_dev(lrb_ctx *c, const uint32_t *d_codes, const uint32_t *d_planes,
                                    const uint64_t *d_code_off, const uint64_t *d_mask_off,
                                    const uint32_t *d_lens, uint64_t n, int mode,
                                    uint32_t *d_counts)
{
    ARG_TRY(c != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    ARG_TRY(mode >= 0 && mode <= 2);
    if (n == 0) return LRB_OK;
    ARG_TRY(d_lens && d_counts);
    if (mode == 1 || d_planes == nullptr) {
        ARG_TRY(d_codes && d_code_off);
        return lrb_kmer_counts_dev(c, d_codes, d_code_off, d_lens, n, 3, d_counts);
    }
    // (mode 2 / planes given: the wave-per-read bit-plane kernel went in round 5 -- the lane-per-read form on the
    // group-transposed planes, lrb_kmer_counts3t_dev, is the k = 3 kernel; per-read layouts are tallied from the codes)
    ARG_TRY(d_codes && d_code_off);
    return lrb_kmer_counts_dev(c, d_codes, d_code_off, d_lens, n, 3, d_counts);
}

// Groups for the lane-per-read kernel.  order (optional, n entries) receives the reads
// sorted by block count (stable), so that the 64 reads of a group are of like length and
// no lane idles while a longer neighbour finishes; without it reads are grouped as given.
static int t_layout(const uint32_t *lens, uint64_t n, uint32_t *order, uint64_t *group_off, int shift)
{
    ARG_TRY(group_off != nullptr && (n == 0 || lens != nullptr));
    ARG_TRY(n <= 0xFFFFFFFFull);
    const uint32_t rnd = (1u << shift) - 1u;
    auto units = [&](uint32_t L) { return (uint32_t)(((uint64_t)L + rnd) >> shift); };
    if (order) {
        // counting sort on the row count (usually a few hundred distinct values)
        uint32_t maxb = 0;
        for (uint64_t r = 0; r < n; ++r) {
            const uint32_t nb = units(lens[r]);
            if (nb > maxb) maxb = nb;
        }
        uint64_t *start = (uint64_t *)calloc((size_t)maxb + 2, sizeof(uint64_t));
        if (!start) return LRB_ERR_NOMEM;
        for (uint64_t r = 0; r < n; ++r) start[units(lens[r]) + 1]++;
        for (uint32_t b = 0; b <= maxb; ++b) start[b + 1] += start[b];
        for (uint64_t r = 0; r < n; ++r) order[start[units(lens[r])]++] = (uint32_t)r;
        free(start);
    }
    const uint64_t ngroups = (n + 63) >> 6;
    uint64_t off = 0;
    for (uint64_t g = 0; g < ngroups; ++g) {
        uint32_t mx = 0;
        for (uint64_t sl = g << 6; sl < n && sl < ((g + 1) << 6); ++sl) {
            const uint32_t nb = units(lens[order ? order[sl] : sl]);
            if (nb > mx) mx = nb;
        }
        group_off[g] = off;
        off += (uint64_t)mx + 1;
    }
    group_off[ngroups] = off;
    return LRB_OK;
}

extern "C" int lrb_planes_t_layout(const uint32_t *lens, uint64_t n, uint32_t *order,
                                   uint64_t *group_off)
{
    return t_layout(lens, n, order, group_off, 5); // rows of 32 bases
}

extern "C" int lrb_codes_t_layout(const uint32_t *lens, uint64_t n, uint32_t *order,
                                  uint64_t *group_off)
{
    return t_layout(lens, n, order, group_off, 6); // rows of 64 bases
}

extern "C" int lrb_codes_t_from_codes_dev(lrb_ctx *c, const uint32_t *d_codes, const uint64_t *d_code_off,
                                          const uint64_t *d_group_off, const uint32_t *d_order, uint64_t n,
                                          uint32_t *d_codes_t)
{
    ARG_TRY(c != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    if (n == 0) return LRB_OK;
    ARG_TRY(d_codes && d_code_off && d_group_off && d_codes_t);
    const uint64_t ngroups = (n + 63) >> 6;
    ARG_TRY(ngroups <= 0x7FFFFFFFull);
    hipLaunchKernelGGL(codes_t_kernel, dim3((unsigned)ngroups), dim3(256), 0, c->stream, d_codes, d_code_off,
                       d_group_off, d_order, n, reinterpret_cast<uint4 *>(d_codes_t));
    HIP_TRY(hipGetLastError());
    return LRB_OK;
}

static int k1_lane_launch(lrb_ctx *c, int k, const uint32_t *d_codes_t, const uint64_t *d_group_off,
                          const uint32_t *d_order, const uint32_t *d_lens, uint64_t n, uint32_t *d_counts)
{
    ARG_TRY(c != nullptr && (k == 3 || k == 4 || k == 5));
    HIP_TRY(hipSetDevice(c->device));
    if (n == 0) return LRB_OK;
    ARG_TRY(d_codes_t && d_group_off && d_lens && d_counts);
    const uint64_t ngroups = (n + 63) >> 6;
    ARG_TRY(ngroups <= 0x7FFFFFFFull);
    const uint4 *ct = reinterpret_cast<const uint4 *>(d_codes_t);
    if (k == 3) {
        // k = 3 on this layout too (the default for k = 3 is the bit-plane kernel on lrb_planes_t_*, 0.736 ms per 1 M
        // reads of 10 kb; this one 0.728): 4-mers at even positions into a 256-bin half-group histogram (16 KB), the
        // flush folds them into the 32 classes of 3-mers -- one LDS atomic per TWO windows
        constexpr size_t smem3 = 256 * 64 + 1024;
        ARG_TRY(ngroups <= 0x3FFFFFFFull);
        hipLaunchKernelGGL((k1_lane4s2_kernel<8, 2, false, 3>), dim3((unsigned)(2 * ngroups)), dim3(512), smem3, c->stream, ct,
                           d_group_off, d_order, d_lens, n, d_counts, (uint64_t *)nullptr);
        HIP_TRY(hipGetLastError());
        return LRB_OK;
    }
    // one group of 64 reads (k = 5: half a group) per workgroup, its waves sharing the histogram: k = 4
    // 32 KB and 4 waves (five workgroups to a CU), k = 5 64 KB and 8 waves (two to a CU); the dispatcher
    // hands a CU the next group as soon as one retires
    if (k == 4) {
        // 5-mers at even positions, half the tallies: half groups, 64 KB + tail words, eight waves, two to a CU; a workgroup
        // per half group -- the dispatcher hands a CU the next one as soon as one retires (two resident workgroups per CU
        // walking everything measured slower: they fall into step and flush together)
        constexpr size_t smem = 65536 + 1024;   // histogram + tail / output row / class tables of the flush
        static lrb_per_device_once attr_s2;
        if (attr_s2.need(c->device)) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k1_lane4s2_kernel<8, 2>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        }
        ARG_TRY(ngroups <= 0x3FFFFFFFull);
        hipLaunchKernelGGL((k1_lane4s2_kernel<8, 2>), dim3((unsigned)(2 * ngroups)), dim3(512), smem, c->stream, ct,
                           d_group_off, d_order, d_lens, n, d_counts);
    } else {
        // k = 5: two half-groups per CU, 2 x (64 KB, eight waves) -- one's flush (2 KB of output per read) overlaps
        // the other's tally; a whole group per CU (128 KB, sixteen waves) measured 1.98 ms against 1.61 ms per 1 M reads
        static lrb_per_device_once attr_set;
        if (attr_set.need(c->device)) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k1_lane4_kernel<5, 8, 2, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 2560));
        }
        ARG_TRY(ngroups <= 0x3FFFFFFFull);
        // 64 KB of histogram + the flush's tables (output row per column 256 B, class table 2 KB)
        hipLaunchKernelGGL((k1_lane4_kernel<5, 8, 2, true>), dim3((unsigned)(2 * ngroups)), dim3(512), 65536 + 2560, c->stream, ct,
                           d_group_off, d_order, d_lens, n, d_counts);
    }
    HIP_TRY(hipGetLastError());
    return LRB_OK;
}

extern "C" int lrb_kmer_counts4t_dev(lrb_ctx *c, const uint32_t *d_codes_t, const uint64_t *d_group_off,
                                     const uint32_t *d_order, const uint32_t *d_lens, uint64_t n,
                                     uint32_t *d_counts)
{
    return k1_lane_launch(c, 4, d_codes_t, d_group_off, d_order, d_lens, n, d_counts);
}

extern "C" int lrb_kmer_counts_t_dev(lrb_ctx *c, int k, const uint32_t *d_codes_t, const uint64_t *d_group_off,
                                     const uint32_t *d_order, const uint32_t *d_lens, uint64_t n,
                                     uint32_t *d_counts)
{
    return k1_lane_launch(c, k, d_codes_t, d_group_off, d_order, d_lens, n, d_counts);
}

extern "C" int lrb_planes_t_from_planes_dev(lrb_ctx *c, const uint32_t *d_planes,
                                            const uint64_t *d_mask_off, const uint64_t *d_group_off,
                                            const uint32_t *d_order, uint64_t n, uint32_t *d_planes_t)
{
    ARG_TRY(c != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    if (n == 0) return LRB_OK;
    ARG_TRY(d_planes && d_mask_off && d_group_off && d_planes_t);
    const uint64_t ngroups = (n + 63) >> 6;
    ARG_TRY(ngroups <= 0x7FFFFFFFull);
    hipLaunchKernelGGL(planes_t_kernel, dim3((unsigned)ngroups), dim3(256), 0, c->stream, d_planes,
                       d_mask_off, d_group_off, d_order, n, reinterpret_cast<uint2 *>(d_planes_t));
    HIP_TRY(hipGetLastError());
    return LRB_OK;
}

extern "C" int lrb_pack_planes_t_dev(lrb_ctx *c, const uint8_t *d_seqs, const uint64_t *d_offs,
                                     const uint64_t *d_group_off, const uint32_t *d_order, uint64_t n,
                                     uint32_t *d_planes_t)
{
    ARG_TRY(c != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    if (n == 0) return LRB_OK;
    ARG_TRY(d_seqs && d_offs && d_group_off && d_planes_t);
    const uint64_t ngroups = (n + 63) >> 6;
    ARG_TRY(ngroups <= 0x7FFFFFFFull);
    hipLaunchKernelGGL(pack_planes_t_kernel, dim3((unsigned)ngroups), dim3(256), 0, c->stream, d_seqs,
                       d_offs, d_group_off, d_order, n, reinterpret_cast<uint2 *>(d_planes_t));
    HIP_TRY(hipGetLastError());
    return LRB_OK;
}

extern "C" int lrb_kmer_counts3t_dev(lrb_ctx *c, const uint32_t *d_planes_t,
                                     const uint64_t *d_group_off, const uint32_t *d_order,
                                     const uint32_t *d_lens, uint64_t n, uint32_t *d_counts)
{
    ARG_TRY(c != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    if (n == 0) return LRB_OK;
    ARG_TRY(d_planes_t && d_group_off && d_lens && d_counts);
    const uint64_t ngroups = (n + 63) >> 6;
    // one group per wave, as many workgroups as that takes: the dispatcher refills a CU when a
    // workgroup retires, which balances the CUs at whole-group granularity (a fixed grid of
    // resident waves looping over groups left CUs with 64 vs 60 groups at 1 M reads)
    uint64_t blocks = (ngroups + 3) / 4;
    if (blocks > 0x7FFFFFFFull) blocks = 0x7FFFFFFFull;
    const int grid = (int)blocks;
    hipLaunchKernelGGL(k1_swar3_lane_kernel, dim3(grid), dim3(256), 0, c->stream,
                       reinterpret_cast<const uint2 *>(d_planes_t), d_group_off, d_order, d_lens, n,
                       d_counts);
    HIP_TRY(hipGetLastError());
    return LRB_OK;
}

// ---- K2 --------------------------------------------------------------------
extern "C" int lrb_k15_accumulate_dev(lrb_ctx *c, const uint32_t *d_codes, const uint32_t *d_mask,
                                      const uint64_t *d_code_off, const uint64_t *d_mask_off,
                                      const uint32_t *d_lens, uint64_t n, uint32_t *d_table)
{
    ARG_TRY(c != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    if (n == 0) return LRB_OK;
    ARG_TRY(d_codes && d_mask && d_code_off && d_mask_off && d_lens && d_table);
    const int grid = grid_for_waves(c, n, 4, 8);
    hipLaunchKernelGGL(k15_accum_kernel, dim3(grid), dim3(256), 0, c->stream, d_codes, d_mask,
                       d_code_off, d_mask_off, d_lens, n, d_table);
    HIP_TRY(hipGetLastError());
    return LRB_OK;
}

// (The forward-table partition route of rounds 1-3 -- count / scan / part1 / part2 / slice kernels -- went in round 5: the
// product tallies into the CANONICAL HALF of the table from window lists, lrb_lists.hip / lrb_packed_k15_tally_half_many.
// The exported name stays for callers that want forward tallies; it is the direct kernel.)
extern "C" int lrb_k15_accumulate_part_dev(lrb_ctx *c, const uint32_t *d_codes, const uint32_t *d_mask,
                                           const uint64_t *d_code_off, const uint64_t *d_mask_off,
                                           const uint32_t *d_lens, uint64_t n, uint64_t max_windows,
                                           uint32_t *d_table)
{
    (void)max_windows;
    return lrb_k15_accumulate_dev(c, d_codes, d_mask, d_code_off, d_mask_off, d_lens, n, d_table);
}

extern "C" int lrb_k15_mirror_dev(lrb_ctx *c, uint32_t *d_table)
{
    ARG_TRY(c != nullptr && d_table != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    hipLaunchKernelGGL(k15_mirror_kernel, dim3(1u << 18), dim3(256), 0, c->stream, d_table);
    HIP_TRY(hipGetLastError());
    return LRB_OK;
}

extern "C" int lrb_k15_fold_half_dev(lrb_ctx *c, const uint32_t *d_table, uint32_t *d_half)
{
    ARG_TRY(c != nullptr && d_table != nullptr && d_half != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    hipLaunchKernelGGL(k15_fold_half_kernel, dim3(1u << 17), dim3(256), 0, c->stream, d_table, d_half);
    HIP_TRY(hipGetLastError());
    return LRB_OK;
}

extern "C" int lrb_k15_expand_half_dev(lrb_ctx *c, const uint32_t *d_half, uint32_t *d_table)
{
    ARG_TRY(c != nullptr && d_table != nullptr && d_half != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    hipLaunchKernelGGL(k15_expand_half_kernel, dim3(1u << 17), dim3(256), 0, c->stream, d_half, d_table);
    HIP_TRY(hipGetLastError());
    return LRB_OK;
}

// ---- K3 --------------------------------------------------------------------
extern "C" int lrb_cov_hist_dev(lrb_ctx *c, const uint32_t *d_codes, const uint32_t *d_mask,
                                const uint64_t *d_code_off, const uint64_t *d_mask_off,
                                const uint32_t *d_lens, uint64_t n, const uint32_t *d_table,
                                int64_t bin_size, int bins, uint32_t *d_hist, uint32_t *d_sums)
{
    ARG_TRY(c != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    ARG_TRY(bin_size >= 1);
    ARG_TRY(bins >= 1 && bins <= 1024);
    if (n == 0) return LRB_OK;
    ARG_TRY(d_codes && d_mask && d_code_off && d_mask_off && d_lens && d_table && d_hist && d_sums);
    // a bin width beyond uint32 puts every count in bin 0, same as 0xFFFFFFFF
    const uint32_t bs = bin_size > 0xFFFFFFFFll ? 0xFFFFFFFFu : (uint32_t)bin_size;
    uint32_t sub_log2 = 5;
    while (sub_log2 > 0 && ((uint32_t)bins << sub_log2) > 4096) --sub_log2;
    const size_t smem = (size_t)4 * ((uint32_t)bins << sub_log2) * 4;
    int per_cu = (int)((160 * 1024) / (smem ? smem : 1));
    if (per_cu > 8) per_cu = 8;
    const int grid = grid_for_waves(c, n, 4, per_cu);
    hipLaunchKernelGGL(cov_hist_kernel, dim3(grid), dim3(256), smem, c->stream, d_codes, d_mask,
                       d_code_off, d_mask_off, d_lens, n, d_table, bs, (uint32_t)bins, sub_log2,
                       d_hist, d_sums);
    HIP_TRY(hipGetLastError());
    return LRB_OK;
}

extern "C" int lrb_cov_map_build_dev(lrb_ctx *c, const uint32_t *d_table, int64_t bin_size, int bins, uint8_t *d_map)
{
    ARG_TRY(c != nullptr && d_table != nullptr && d_map != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    ARG_TRY(bin_size >= 1);
    ARG_TRY(bins >= 1 && bins <= 256); // a bin id is one byte of the map
    const uint32_t bs = bin_size > 0xFFFFFFFFll ? 0xFFFFFFFFu : (uint32_t)bin_size;
    hipLaunchKernelGGL(cov_map_build_kernel, dim3(c->n_cu * 32), dim3(256), 0, c->stream, d_table, bs, (uint32_t)bins,
                       d_map);
    HIP_TRY(hipGetLastError());
    return LRB_OK;
}

extern "C" int lrb_cov_hist_map_dev(lrb_ctx *c, const uint32_t *d_codes, const uint32_t *d_mask,
                                    const uint64_t *d_code_off, const uint64_t *d_mask_off,
                                    const uint32_t *d_lens, uint64_t n, const uint8_t *d_map, int bins,
                                    uint32_t *d_hist, uint32_t *d_sums)
{
    ARG_TRY(c != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    ARG_TRY(bins >= 1 && bins <= 256);
    if (n == 0) return LRB_OK;
    ARG_TRY(d_codes && d_mask && d_code_off && d_mask_off && d_lens && d_map && d_hist && d_sums);
    uint32_t sub_log2 = 5;
    while (sub_log2 > 0 && ((uint32_t)bins << sub_log2) > 4096) --sub_log2;
    const size_t smem = (size_t)4 * ((uint32_t)bins << sub_log2) * 4;
    int per_cu = (int)((160 * 1024) / (smem ? smem : 1));
    if (per_cu > 8) per_cu = 8;
    const int grid = grid_for_waves(c, n, 4, per_cu);
    hipLaunchKernelGGL(cov_hist_map_kernel, dim3(grid), dim3(256), smem, c->stream, d_codes, d_mask, d_code_off,
                       d_mask_off, d_lens, n, d_map, (uint32_t)bins, sub_log2, 0u, d_hist, d_sums);
    HIP_TRY(hipGetLastError());
    return LRB_OK;
}

// reads of more than 65,535 windows are not in the window lists (a u16 counter could overflow): the gather kernel
// tallies them
int lrb_cov_hist_map_long(lrb_ctx *c, const uint32_t *d_codes, const uint32_t *d_mask, const uint64_t *d_code_off,
                          const uint64_t *d_mask_off, const uint32_t *d_lens, uint64_t n, const uint8_t *d_map,
                          int bins, uint32_t *d_hist, uint32_t *d_sums)
{
    uint32_t sub_log2 = 5;
    while (sub_log2 > 0 && ((uint32_t)bins << sub_log2) > 4096) --sub_log2;
    const size_t gs = (size_t)4 * ((uint32_t)bins << sub_log2) * 4;
    int per_cu = (int)((160 * 1024) / (gs ? gs : 1));
    if (per_cu > 8) per_cu = 8;
    const int grid = grid_for_waves(c, n, 4, per_cu);
    hipLaunchKernelGGL(cov_hist_map_kernel, dim3(grid), dim3(256), gs, c->stream, d_codes, d_mask, d_code_off,
                       d_mask_off, d_lens, n, d_map, (uint32_t)bins, sub_log2, CJ_MAX_WINDOWS + 15u, d_hist, d_sums);
    HIP_TRY(hipGetLastError());
    return LRB_OK;
}

// ... and one atomic per window into the canonical half
int lrb_k15_accum_half_long(lrb_ctx *c, const uint32_t *d_codes, const uint32_t *d_mask, const uint64_t *d_code_off,
                            const uint64_t *d_mask_off, const uint32_t *d_lens, uint64_t n, uint32_t min_len,
                            uint32_t *d_half)
{
    hipLaunchKernelGGL(k15_accum_half_kernel, dim3(grid_for_waves(c, n, 4, 8)), dim3(256), 0, c->stream, d_codes, d_mask,
                       d_code_off, d_mask_off, d_lens, n, min_len, d_half);
    HIP_TRY(hipGetLastError());
    return LRB_OK;
}

extern "C" int lrb_k15_accumulate_half_dev(lrb_ctx *c, const uint32_t *d_codes, const uint32_t *d_mask,
                                           const uint64_t *d_code_off, const uint64_t *d_mask_off,
                                           const uint32_t *d_lens, uint64_t n, uint32_t *d_half)
{
    ARG_TRY(c != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    if (n == 0) return LRB_OK;
    ARG_TRY(d_codes && d_mask && d_code_off && d_mask_off && d_lens && d_half);
    hipLaunchKernelGGL(k15_accum_half_kernel, dim3(grid_for_waves(c, n, 4, 8)), dim3(256), 0, c->stream, d_codes, d_mask,
                       d_code_off, d_mask_off, d_lens, n, 0u, d_half);
    HIP_TRY(hipGetLastError());
    return LRB_OK;
}

extern "C" int lrb_cov_map_build_half_dev(lrb_ctx *c, const uint32_t *d_half, int64_t bin_size, int bins, uint8_t *d_map)
{
    ARG_TRY(c != nullptr && d_half != nullptr && d_map != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    ARG_TRY(bin_size >= 1);
    ARG_TRY(bins >= 1 && bins <= 256);
    const uint32_t bs = bin_size > 0xFFFFFFFFll ? 0xFFFFFFFFu : (uint32_t)bin_size;
    hipLaunchKernelGGL(cov_map_build_half_kernel, dim3(c->n_cu * 32), dim3(256), 0, c->stream, d_half, bs, (uint32_t)bins,
                       d_map);
    HIP_TRY(hipGetLastError());
    return LRB_OK;
}

// ---- host-pointer convenience paths ----------------------------------------
int lrb_ws_get(lrb_ctx *c, int slot, uint64_t bytes, void **p)
{
    if (slot == 8 || (slot >= 11 && slot <= 15)) ++c->lists_epoch;
    if (c->ws_bytes[slot] < bytes) {
        if (c->ws[slot]) HIP_TRY(hipFree(c->ws[slot]));
        c->ws[slot] = nullptr;
        c->ws_bytes[slot] = 0;
        // small workspaces grow by a quarter beyond the request (fewer re-allocations while batches vary); from 1 GiB
        // on the request is taken as it stands -- the partition buffers and slice lists are sized from what is free
        // (half of it), and a quarter on top of that budget is what the caller did not plan for
        uint64_t want = bytes + (bytes < (1ull << 30) ? (bytes >> 2) : 0) + 4096;
        if (hipMalloc(&c->ws[slot], want) != hipSuccess) {
            (void)hipGetLastError();
            want = bytes + 4096;
            HIP_TRY(hipMalloc(&c->ws[slot], want));
        }
        c->ws_bytes[slot] = want;
    }
    *p = c->ws[slot];
    return LRB_OK;
}

struct packed_dev {
    uint32_t *codes, *mask, *lens, *planes;
    uint64_t *code_off, *mask_off;
    // k = 3 fast path: group-transposed bit planes
    uint32_t *planes_t, *order;
    uint64_t *group_off;
    // k = 4, 5 fast path: group-transposed 2-bit codes
    uint32_t *codes_t, *order4;
    uint64_t *group_off4;
};

// H2D + pack into the context workspace (slots 0..5).  Synchronous on return of
// the H2D copies only; the pack kernel is left enqueued.
struct lrb_packed {
    packed_dev pd;
    void *owned[8]; // offsets(3 arrays), lens, codes, mask, planes_t, order+group_off, codes_t, order4+group_off4
    uint64_t n, bytes, total_bases;
    uint64_t code_words, mask_words; // sizes of pd.codes / pd.mask in uint32 words
    bool has_planes, has_codes_t;
};

// One resident batch in a table of batches (device copy): what the many-batch kernels below reach it by.
struct pack_desc {
    const uint32_t *mask, *lens, *order4;
    const uint64_t *code_off, *mask_off, *group_off4;
    uint64_t n, mask_words;
    uint64_t read0, mask0;      // first read / first mask word of this batch in the merged arrays
    uint64_t code_delta;        // words from the common codes base to this batch's codes
    uint64_t group0, row_delta; // (K1) first merged group; 1-KiB rows from the common codes_t base to this batch's
    uint32_t last;              // the batch that writes the merged offsets' closing entry
};

static int add_transposed_layout(lrb_ctx *c, const uint64_t *offs, uint64_t n, int which, void *d_seqs, void *d_offs,
                                 packed_dev *pd, lrb_packed *own);

// layouts: bit 0 group-transposed bit planes (k = 3), bit 1 group-transposed codes (k = 4, 5)
// seqs_on_device: `seqs` is the bases' address in HBM (byte offs[0] of the caller's buffer is at seqs + offs[0], as for a host
// buffer): nothing but the small offset / length arrays crosses PCIe
static int upload_and_pack(lrb_ctx *c, const uint8_t *seqs, const uint64_t *offs, uint64_t n,
                           bool want_mask, int layouts, packed_dev *pd,
                           lrb_packed *own = nullptr, bool seqs_on_device = false)
{
    const bool want_planes_t = (layouts & 1) != 0, want_codes_t = (layouts & 2) != 0;
    HIP_TRY(hipSetDevice(c->device));
    uint64_t *h_code_off = (uint64_t *)malloc(sizeof(uint64_t) * (n + 1) * 2);
    uint32_t *h_lens = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    if (!h_code_off || !h_lens) {
        free(h_code_off);
        free(h_lens);
        lrb_set_error("host allocation failed%s%s", "", "");
        return LRB_ERR_NOMEM;
    }
    uint64_t *h_mask_off = h_code_off + (n + 1);
    int rc = lrb_pack_layout(offs, n, h_lens, h_code_off, h_mask_off);
    if (rc != LRB_OK) {
        free(h_code_off);
        free(h_lens);
        return rc;
    }
    const uint64_t seq_bytes = offs[n] - offs[0];
    void *d_seqs, *d_offs, *d_co, *d_mo, *d_lens, *d_codes, *d_mask = nullptr;
    void *d_planes = nullptr; // per-read planes are a device-level format only
#define WS_TRY(x)                 \
    do {                          \
        int rc_ = (x);            \
        if (rc_ != LRB_OK) {      \
            free(h_code_off);     \
            free(h_lens);         \
            return rc_;           \
        }                         \
    } while (0)
    if (seqs_on_device)
        d_seqs = (void *)(seqs + offs[0]);
    else
        WS_TRY(ws_get(c, 0, seq_bytes + 64, &d_seqs));
    if (!own) {
        WS_TRY(ws_get(c, 1, sizeof(uint64_t) * (n + 1) * 3, &d_offs));
        WS_TRY(ws_get(c, 2, sizeof(uint32_t) * n, &d_lens));
        WS_TRY(ws_get(c, 3, sizeof(uint32_t) * h_code_off[n], &d_codes));
        if (want_mask) WS_TRY(ws_get(c, 4, sizeof(uint32_t) * h_mask_off[n], &d_mask));
    } else {
        // buffers that outlive the call: the batch stays resident in HBM
        const uint64_t b_off = sizeof(uint64_t) * (n + 1) * 3, b_len = sizeof(uint32_t) * n + 16;
        const uint64_t b_codes = sizeof(uint32_t) * h_code_off[n];
        const uint64_t b_mask = (want_mask ? sizeof(uint32_t) * h_mask_off[n] : 0) + 16;
        hipError_t ea = hipMalloc(&own->owned[0], b_off);
        if (ea == hipSuccess) ea = hipMalloc(&own->owned[1], b_len);
        if (ea == hipSuccess) ea = hipMalloc(&own->owned[2], b_codes);
        if (ea == hipSuccess) ea = hipMalloc(&own->owned[3], b_mask);
        if (ea != hipSuccess) {
            lrb_set_error("device allocation for a resident batch failed: %s%s", hipGetErrorString(ea), "");
            free(h_code_off);
            free(h_lens);
            return LRB_ERR_NOMEM;
        }
        d_offs = own->owned[0];
        d_lens = own->owned[1];
        d_codes = own->owned[2];
        if (want_mask) d_mask = own->owned[3];
        own->bytes = b_off + b_len + b_codes + b_mask;
        own->code_words = h_code_off[n];
        own->mask_words = want_mask ? h_mask_off[n] : 0;
    }
    d_co = (uint64_t *)d_offs + (n + 1);
    d_mo = (uint64_t *)d_offs + 2 * (n + 1);
#undef WS_TRY
    hipError_t e = hipSuccess;
    // offsets are rebased to offs[0] on the device side
    uint64_t *h_offs0 = (uint64_t *)malloc(sizeof(uint64_t) * (n + 1));
    if (!h_offs0) {
        free(h_code_off);
        free(h_lens);
        return LRB_ERR_NOMEM;
    }
    for (uint64_t i = 0; i <= n; ++i) h_offs0[i] = offs[i] - offs[0];
    if (seq_bytes && !seqs_on_device)
        e = hipMemcpyAsync(d_seqs, seqs + offs[0], seq_bytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(d_offs, h_offs0, sizeof(uint64_t) * (n + 1), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(d_co, h_code_off, sizeof(uint64_t) * (n + 1), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(d_mo, h_mask_off, sizeof(uint64_t) * (n + 1), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(d_lens, h_lens, sizeof(uint32_t) * n, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    free(h_offs0);
    free(h_code_off);
    free(h_lens);
    if (e != hipSuccess) {
        lrb_set_error("upload failed: %s%s", hipGetErrorString(e), "");
        return LRB_ERR_HIP;
    }
    pd->codes = (uint32_t *)d_codes;
    pd->mask = (uint32_t *)d_mask;
    pd->planes = (uint32_t *)d_planes;
    pd->lens = (uint32_t *)d_lens;
    pd->code_off = (uint64_t *)d_co;
    pd->mask_off = (uint64_t *)d_mo;
    pd->planes_t = nullptr;
    pd->order = nullptr;
    pd->group_off = nullptr;
    pd->codes_t = nullptr;
    pd->order4 = nullptr;
    pd->group_off4 = nullptr;
    rc = lrb_pack_reads_dev(c, (const uint8_t *)d_seqs, seq_bytes, (const uint64_t *)d_offs, n,
                            pd->code_off, pd->mask_off, pd->codes, pd->mask, pd->planes);
    if (rc != LRB_OK) return rc;
    if (want_codes_t) {
        rc = add_transposed_layout(c, offs, n, 2, d_seqs, d_offs, pd, own);
        if (rc != LRB_OK) return rc;
    }
    if (!want_planes_t) return rc;
    return add_transposed_layout(c, offs, n, 1, d_seqs, d_offs, pd, own);
}

// Length-sorted groups of 64 reads and the group-transposed form the lane-per-read kernels read:
// which = 1 bit planes written straight from ASCII (k = 3), which = 2 2-bit codes from pd->codes (k = 4, 5).
static int add_transposed_layout(lrb_ctx *c, const uint64_t *offs, uint64_t n, int which, void *d_seqs, void *d_offs,
                                 packed_dev *pd, lrb_packed *own)
{
    int rc;
    const uint64_t ngroups = (n + 63) >> 6;
    uint32_t *h_order = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    uint64_t *h_goff = (uint64_t *)malloc(sizeof(uint64_t) * (ngroups + 1));
    uint32_t *h_lens2 = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    if (!h_order || !h_goff || !h_lens2) {
        free(h_order);
        free(h_goff);
        free(h_lens2);
        lrb_set_error("host allocation failed%s%s", "", "");
        return LRB_ERR_NOMEM;
    }
    for (uint64_t i = 0; i < n; ++i) h_lens2[i] = (uint32_t)(offs[i + 1] - offs[i]);
    rc = which == 1 ? lrb_planes_t_layout(h_lens2, n, h_order, h_goff) : lrb_codes_t_layout(h_lens2, n, h_order, h_goff);
    free(h_lens2);
    void *d_pt = nullptr, *d_og = nullptr;
    const uint64_t b_pt = sizeof(uint32_t) * (which == 1 ? 128 : 256) * h_goff[ngroups] + 16;
    const uint64_t b_og = sizeof(uint32_t) * n + sizeof(uint64_t) * (ngroups + 1) + 32;
    if (rc == LRB_OK) {
        if (own) {
            const int s0 = which == 1 ? 4 : 6;
            hipError_t ea = hipMalloc(&own->owned[s0], b_pt);
            if (ea == hipSuccess) ea = hipMalloc(&own->owned[s0 + 1], b_og);
            if (ea != hipSuccess) {
                lrb_set_error("device allocation for a resident batch failed: %s%s", hipGetErrorString(ea), "");
                rc = LRB_ERR_NOMEM;
            }
            d_pt = own->owned[s0];
            d_og = own->owned[s0 + 1];
            if (rc == LRB_OK) own->bytes += b_pt + b_og;
        } else {
            rc = ws_get(c, which == 1 ? 7 : 8, b_pt, &d_pt);
            // order + group_off: the mask slot is free on the composition-only path (k = 3); k = 4, 5 take their own
            if (rc == LRB_OK) rc = ws_get(c, which == 1 ? 4 : 9, b_og, &d_og);
        }
    }
    if (rc == LRB_OK) {
        // group_off first (8-byte aligned), then order
        uint64_t *d_goff = (uint64_t *)d_og;
        uint32_t *d_order = (uint32_t *)(d_goff + ngroups + 1);
        hipError_t e2 = hipMemcpyAsync(d_goff, h_goff, sizeof(uint64_t) * (ngroups + 1), hipMemcpyHostToDevice, c->stream);
        if (e2 == hipSuccess && n)
            e2 = hipMemcpyAsync(d_order, h_order, sizeof(uint32_t) * n, hipMemcpyHostToDevice, c->stream);
        if (e2 == hipSuccess) e2 = hipStreamSynchronize(c->stream);
        if (e2 != hipSuccess) {
            lrb_set_error("upload failed: %s%s", hipGetErrorString(e2), "");
            rc = LRB_ERR_HIP;
        } else if (which == 1 && d_seqs == nullptr) {
            // a batch that arrived packed (lrb_packed_create_packed): the planes of the reads from their codes, in
            // workspace, then transposed
            pd->planes_t = (uint32_t *)d_pt;
            pd->order = d_order;
            pd->group_off = d_goff;
            void *d_pl;
            rc = ws_get(c, 1, sizeof(uint32_t) * 2 * (own ? own->mask_words : 0) + 64, &d_pl);
            if (rc == LRB_OK) rc = lrb_planes_from_codes_dev(c, pd->codes, pd->code_off, pd->mask_off, n, (uint32_t *)d_pl);
            if (rc == LRB_OK)
                rc = lrb_planes_t_from_planes_dev(c, (const uint32_t *)d_pl, pd->mask_off, pd->group_off, pd->order, n, pd->planes_t);
        } else if (which == 1) {
            pd->planes_t = (uint32_t *)d_pt;
            pd->order = d_order;
            pd->group_off = d_goff;
            rc = lrb_pack_planes_t_dev(c, (const uint8_t *)d_seqs, (const uint64_t *)d_offs, pd->group_off,
                                       pd->order, n, pd->planes_t);
        } else {
            pd->codes_t = (uint32_t *)d_pt;
            pd->order4 = d_order;
            pd->group_off4 = d_goff;
            rc = lrb_codes_t_from_codes_dev(c, pd->codes, pd->code_off, pd->group_off4, pd->order4, n, pd->codes_t);
        }
    }
    free(h_order);
    free(h_goff);
    return rc;
}

extern "C" int lrb_kmer_counts_host(lrb_ctx *c, const uint8_t *seqs, const uint64_t *offs,
                                    uint64_t n, int k, uint32_t *counts)
{
    ARG_TRY(c != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    ARG_TRY(k >= 3 && k <= 5);
    if (n == 0) return LRB_OK;
    ARG_TRY(seqs && offs && counts);
    packed_dev pd;
    int rc = upload_and_pack(c, seqs, offs, n, false, k == 3 ? 1 : 2, &pd);
    if (rc != LRB_OK) return rc;
    void *d_counts;
    const uint64_t bytes = sizeof(uint32_t) * n * c->dim[k];
    rc = ws_get(c, 5, bytes, &d_counts);
    if (rc != LRB_OK) return rc;
    if (k == 3)
        rc = lrb_kmer_counts3t_dev(c, pd.planes_t, pd.group_off, pd.order, pd.lens, n,
                                   (uint32_t *)d_counts);
    else
        rc = lrb_kmer_counts_t_dev(c, k, pd.codes_t, pd.group_off4, pd.order4, pd.lens, n, (uint32_t *)d_counts);
    if (rc != LRB_OK) return rc;
    return lrb_copy_d2h(c, counts, d_counts, bytes);
}

extern "C" int lrb_k15_accumulate_host(lrb_ctx *c, const uint8_t *seqs, const uint64_t *offs,
                                       uint64_t n, uint32_t *d_table)
{
    ARG_TRY(c != nullptr && d_table != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    if (n == 0) return LRB_OK;
    ARG_TRY(seqs && offs);
    packed_dev pd;
    int rc = upload_and_pack(c, seqs, offs, n, true, 0, &pd);
    if (rc != LRB_OK) return rc;
    rc = lrb_k15_accumulate_part_dev(c, pd.codes, pd.mask, pd.code_off, pd.mask_off, pd.lens, n,
                                     offs[n] - offs[0], d_table);
    if (rc != LRB_OK) return rc;
    return lrb_ctx_sync(c);
}

extern "C" int lrb_cov_hist_host(lrb_ctx *c, const uint8_t *seqs, const uint64_t *offs, uint64_t n,
                                 const uint32_t *d_table, int64_t bin_size, int bins,
                                 uint32_t *hist, uint32_t *sums)
{
    ARG_TRY(c != nullptr && d_table != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    ARG_TRY(bin_size >= 1);
    ARG_TRY(bins >= 1 && bins <= 1024);
    if (n == 0) return LRB_OK;
    ARG_TRY(seqs && offs && hist && sums);
    packed_dev pd;
    int rc = upload_and_pack(c, seqs, offs, n, true, 0, &pd);
    if (rc != LRB_OK) return rc;
    void *d_hist, *d_sums;
    rc = ws_get(c, 5, sizeof(uint32_t) * n * bins, &d_hist);
    if (rc != LRB_OK) return rc;
    rc = ws_get(c, 6, sizeof(uint32_t) * n, &d_sums);
    if (rc != LRB_OK) return rc;
    rc = lrb_cov_hist_dev(c, pd.codes, pd.mask, pd.code_off, pd.mask_off, pd.lens, n, d_table,
                          bin_size, bins, (uint32_t *)d_hist, (uint32_t *)d_sums);
    if (rc != LRB_OK) return rc;
    rc = lrb_copy_d2h(c, hist, d_hist, sizeof(uint32_t) * n * bins);
    if (rc != LRB_OK) return rc;
    return lrb_copy_d2h(c, sums, d_sums, sizeof(uint32_t) * n);
}

// ---- resident batches --------------------------------------------------------
// A batch of reads uploaded and packed ONCE and kept in HBM, so that the composition,
// table and coverage stages of one run do not parse and upload the file three times.
static int packed_create_from(lrb_ctx *c, const uint8_t *seqs, const uint64_t *offs, uint64_t n, int with_planes,
                              bool seqs_on_device, lrb_packed **out)
{
    ARG_TRY(c != nullptr && out != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    ARG_TRY(n == 0 || (seqs && offs));
    lrb_packed *p = (lrb_packed *)calloc(1, sizeof(lrb_packed));
    if (!p) return LRB_ERR_NOMEM;
    p->n = n;
    p->total_bases = n ? offs[n] - offs[0] : 0;
    p->has_planes = (with_planes & 1) != 0;
    p->has_codes_t = (with_planes & 2) != 0;
    if (n) {
        int rc = upload_and_pack(c, seqs, offs, n, true, with_planes & 3, &p->pd, p, seqs_on_device);
        if (rc == LRB_OK) rc = lrb_ctx_sync(c);
        if (rc != LRB_OK) {
            for (int i = 0; i < 8; ++i)
                if (p->owned[i]) (void)hipFree(p->owned[i]);
            free(p);
            return rc;
        }
    }
    *out = p;
    return LRB_OK;
}

extern "C" int lrb_packed_create(lrb_ctx *c, const uint8_t *seqs, const uint64_t *offs, uint64_t n,
                                 int with_planes, lrb_packed **out)
{
    return packed_create_from(c, seqs, offs, n, with_planes, false, out);
}

extern "C" int lrb_packed_create_dev(lrb_ctx *c, const uint8_t *d_seqs, const uint64_t *offs, uint64_t n,
                                     int with_planes, lrb_packed **out)
{
    return packed_create_from(c, d_seqs, offs, n, with_planes, true, out);
}

// A resident batch from reads that arrive ALREADY PACKED (lrb_pack_reads_host / the parser pool's packed view): the
// codes, masks, offsets and lengths are uploaded as they are -- 0.375 bytes a base over PCIe instead of 1 -- and the
// transposed layouts are made from the codes on the device.  offs: the reads' byte offsets (for the lengths and the
// layouts' length sort), code_off / mask_off as lrb_pack_layout gives them, starting at 0.
extern "C" int lrb_packed_create_packed(lrb_ctx *c, const uint32_t *codes, const uint32_t *mask, const uint64_t *code_off,
                                        const uint64_t *mask_off, const uint32_t *lens, const uint64_t *offs, uint64_t n,
                                        int with_planes, lrb_packed **out)
{
    ARG_TRY(c != nullptr && out != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    ARG_TRY(n == 0 || (codes && mask && code_off && mask_off && lens && offs));
    lrb_packed *p = (lrb_packed *)calloc(1, sizeof(lrb_packed));
    if (!p) return LRB_ERR_NOMEM;
    p->n = n;
    p->total_bases = n ? offs[n] - offs[0] : 0;
    p->has_planes = (with_planes & 1) != 0;
    p->has_codes_t = (with_planes & 2) != 0;
    auto fail = [&](int rc) {
        for (int i = 0; i < 8; ++i)
            if (p->owned[i]) (void)hipFree(p->owned[i]);
        free(p);
        return rc;
    };
    if (n) {
        if (code_off[0] != 0 || mask_off[0] != 0) {
            lrb_set_error("invalid argument: %s%s", "code_off[0] == 0 && mask_off[0] == 0", "");
            return fail(LRB_ERR_ARG);
        }
        const uint64_t b_off = sizeof(uint64_t) * (n + 1) * 3, b_len = sizeof(uint32_t) * n + 16;
        const uint64_t b_codes = sizeof(uint32_t) * code_off[n], b_mask = sizeof(uint32_t) * mask_off[n] + 16;
        hipError_t ea = hipMalloc(&p->owned[0], b_off);
        if (ea == hipSuccess) ea = hipMalloc(&p->owned[1], b_len);
        if (ea == hipSuccess) ea = hipMalloc(&p->owned[2], b_codes);
        if (ea == hipSuccess) ea = hipMalloc(&p->owned[3], b_mask);
        if (ea != hipSuccess) {
            lrb_set_error("device allocation for a resident batch failed: %s%s", hipGetErrorString(ea), "");
            return fail(LRB_ERR_NOMEM);
        }
        p->bytes = b_off + b_len + b_codes + b_mask;
        p->code_words = code_off[n];
        p->mask_words = mask_off[n];
        uint64_t *d_offs = (uint64_t *)p->owned[0], *d_co = d_offs + (n + 1), *d_mo = d_offs + 2 * (n + 1);
        std::vector<uint64_t> offs0(n + 1);
        for (uint64_t i = 0; i <= n; ++i) offs0[i] = offs[i] - offs[0];
        hipError_t e = hipMemcpyAsync(p->owned[2], codes, sizeof(uint32_t) * code_off[n], hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(p->owned[3], mask, sizeof(uint32_t) * mask_off[n], hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_offs, offs0.data(), sizeof(uint64_t) * (n + 1), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_co, code_off, sizeof(uint64_t) * (n + 1), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_mo, mask_off, sizeof(uint64_t) * (n + 1), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(p->owned[1], lens, sizeof(uint32_t) * n, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) {
            lrb_set_error("upload failed: %s%s", hipGetErrorString(e), "");
            return fail(LRB_ERR_HIP);
        }
        packed_dev &pd = p->pd;
        pd.codes = (uint32_t *)p->owned[2];
        pd.mask = (uint32_t *)p->owned[3];
        pd.lens = (uint32_t *)p->owned[1];
        pd.code_off = d_co;
        pd.mask_off = d_mo;
        int rc = LRB_OK;
        if (with_planes & 2) rc = add_transposed_layout(c, offs, n, 2, nullptr, d_offs, &pd, p);
        if (rc == LRB_OK && (with_planes & 1)) rc = add_transposed_layout(c, offs, n, 1, nullptr, d_offs, &pd, p);
        if (rc == LRB_OK) rc = lrb_ctx_sync(c);
        if (rc != LRB_OK) return fail(rc);
    }
    *out = p;
    return LRB_OK;
}

extern "C" int lrb_packed_free(lrb_ctx *c, lrb_packed *p)
{
    ARG_TRY(c != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    if (!p) return LRB_OK;
    (void)hipStreamSynchronize(c->stream);
    lrb_resident_lists_forget_batch(c, p);   // lists in the workspace that reach this batch's codes
    for (int i = 0; i < 8; ++i)
        if (p->owned[i]) (void)hipFree(p->owned[i]);
    free(p);
    return LRB_OK;
}

extern "C" int lrb_packed_info(const lrb_packed *p, uint64_t *n, uint64_t *device_bytes)
{
    ARG_TRY(p != nullptr);
    if (n) *n = p->n;
    if (device_bytes) *device_bytes = p->bytes;
    return LRB_OK;
}

extern "C" int lrb_packed_kmer_counts(lrb_ctx *c, const lrb_packed *p, int k, uint32_t *counts)
{
    ARG_TRY(c != nullptr && p != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    ARG_TRY(k >= 3 && k <= 5);
    if (p->n == 0) return LRB_OK;
    ARG_TRY(counts != nullptr);
    void *d_counts;
    const uint64_t bytes = sizeof(uint32_t) * p->n * c->dim[k];
    int rc = ws_get(c, 5, bytes, &d_counts);
    if (rc != LRB_OK) return rc;
    if (k == 3 && p->has_planes)
        rc = lrb_kmer_counts3t_dev(c, p->pd.planes_t, p->pd.group_off, p->pd.order, p->pd.lens, p->n,
                                   (uint32_t *)d_counts);
    else if (k != 3 && p->has_codes_t)
        rc = lrb_kmer_counts_t_dev(c, k, p->pd.codes_t, p->pd.group_off4, p->pd.order4, p->pd.lens, p->n,
                                   (uint32_t *)d_counts);
    else
        rc = lrb_kmer_counts_dev(c, p->pd.codes, p->pd.code_off, p->pd.lens, p->n, k,
                                 (uint32_t *)d_counts);
    if (rc != LRB_OK) return rc;
    return lrb_copy_d2h(c, counts, d_counts, bytes);
}

// the kernel half of lrb_packed_kmer_counts / lrb_packed_kmer_text: the tallies stay in HBM (d_counts: n x dim uint32)
extern "C" int lrb_packed_kmer_counts_dev(lrb_ctx *c, const lrb_packed *p, int k, uint32_t *d_counts)
{
    ARG_TRY(c != nullptr && p != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    ARG_TRY(k >= 3 && k <= 5);
    if (p->n == 0) return LRB_OK;
    ARG_TRY(d_counts != nullptr);
    if (k == 3 && p->has_planes)
        return lrb_kmer_counts3t_dev(c, p->pd.planes_t, p->pd.group_off, p->pd.order, p->pd.lens, p->n, d_counts);
    if (k != 3 && p->has_codes_t)
        return lrb_kmer_counts_t_dev(c, k, p->pd.codes_t, p->pd.group_off4, p->pd.order4, p->pd.lens, p->n, d_counts);
    return lrb_kmer_counts_dev(c, p->pd.codes, p->pd.code_off, p->pd.lens, p->n, k, d_counts);
}

// ---- K1 of MANY resident batches behind one launch (round 6) ----
// merged (first row, end row) per group, merged order (the read's row in the merged output, ~0 for padding) and merged
// lengths of a table of batches: blockIdx.y = batch
__global__ __launch_bounds__(256) void k1_many_meta_kernel(const pack_desc *__restrict__ descs, uint64_t *__restrict__ group_pairs,
                                                           uint32_t *__restrict__ order_out, uint32_t *__restrict__ lens_out)
{
    const pack_desc d = descs[blockIdx.y];
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t ngroups = (d.n + 63) >> 6;
    for (uint64_t g = t; g < ngroups; g += stride) {
        group_pairs[2 * (d.group0 + g)] = d.group_off4[g] + d.row_delta;
        group_pairs[2 * (d.group0 + g) + 1] = d.group_off4[g + 1] + d.row_delta;
    }
    for (uint64_t sl = t; sl < ngroups * 64; sl += stride)
        order_out[d.group0 * 64 + sl] = sl < d.n ? (uint32_t)((d.order4 ? d.order4[sl] : sl) + d.read0) : 0xFFFFFFFFu;
    for (uint64_t i = t; i < d.n; i += stride) lens_out[d.read0 + i] = d.lens[i];
}

// The k-mer tallies of `count` resident batches, rows in batch order in d_counts (sum of n x dim uint32): for k = 4 on
// batches that hold the group-transposed codes ONE launch of the stride-2 kernel over all their groups (a table of the
// batches, merged group / order / length arrays made by one small kernel: workspace slots 0..3) -- a launch per batch of
// ~6,700 reads is 210 workgroups on a chip that holds 512 and a dispatch latency each, 8.7 ms per 2.5 M reads against
// 2.2 ms of tallying; anything else falls back to lrb_packed_kmer_counts_dev batch by batch.  count-kmers.cpp:66-95.
extern "C" int lrb_packed_kmer_counts_many_dev(lrb_ctx *c, const lrb_packed *const *packs, uint64_t count, int k,
                                               uint32_t *d_counts)
{
    ARG_TRY(c != nullptr && (packs != nullptr || count == 0));
    HIP_TRY(hipSetDevice(c->device));
    ARG_TRY(k >= 3 && k <= 5);
    uint64_t n = 0, groups = 0;
    const uint32_t *base = nullptr;
    bool one_launch = k == 4 && count > 1;
    for (uint64_t i = 0; i < count; ++i) {
        ARG_TRY(packs[i] != nullptr);
        if (packs[i]->n == 0) continue;
        if (!packs[i]->has_codes_t) one_launch = false;
        else if (!base || packs[i]->pd.codes_t < base) base = packs[i]->pd.codes_t;
        n += packs[i]->n;
        groups += (packs[i]->n + 63) >> 6;
    }
    if (n == 0) return LRB_OK;
    ARG_TRY(d_counts != nullptr);
    for (uint64_t i = 0; i < count && one_launch; ++i)   // rows of 1 KiB from the common base, row counts in 32 bits a group
        if (packs[i]->n && ((const char *)packs[i]->pd.codes_t - (const char *)base) % 1024 != 0) one_launch = false;
    if (groups > 0x3FFFFFFFull || n > 0xFFFFFFFEull) one_launch = false;
    if (!one_launch) {
        uint64_t at = 0;
        for (uint64_t i = 0; i < count; ++i) {
            const int rc = lrb_packed_kmer_counts_dev(c, packs[i], k, d_counts + at * c->dim[k]);
            if (rc != LRB_OK) return rc;
            at += packs[i]->n;
        }
        return LRB_OK;
    }
    std::vector<pack_desc> descs;
    descs.reserve(count);
    uint64_t at = 0, g0 = 0;
    for (uint64_t i = 0; i < count; ++i) {
        const lrb_packed *p = packs[i];
        if (p->n == 0) continue;
        pack_desc d = {};
        d.lens = p->pd.lens;
        d.order4 = p->pd.order4;
        d.group_off4 = p->pd.group_off4;
        d.n = p->n;
        d.read0 = at;
        d.group0 = g0;
        d.row_delta = (uint64_t)((const char *)p->pd.codes_t - (const char *)base) / 1024;
        descs.push_back(d);
        at += p->n;
        g0 += (p->n + 63) >> 6;
    }
    void *d_descs, *d_pairs, *d_order, *d_lens;
    int rc = lrb_stage_upload(c, 0, descs.data(), sizeof(pack_desc) * descs.size(), &d_descs);
    if (rc == LRB_OK) rc = ws_get(c, 1, sizeof(uint64_t) * 2 * groups + 64, &d_pairs);
    if (rc == LRB_OK) rc = ws_get(c, 2, sizeof(uint32_t) * n + 64, &d_lens);
    if (rc == LRB_OK) rc = ws_get(c, 3, sizeof(uint32_t) * 64 * groups + 64, &d_order);
    if (rc != LRB_OK) return rc;
    for (size_t d0 = 0; d0 < descs.size(); d0 += 65535) {
        const size_t nd = descs.size() - d0 < 65535 ? descs.size() - d0 : 65535;
        hipLaunchKernelGGL(k1_many_meta_kernel, dim3(8, (unsigned)nd), dim3(256), 0, c->stream, (const pack_desc *)d_descs + d0,
                           (uint64_t *)d_pairs, (uint32_t *)d_order, (uint32_t *)d_lens);
    }
    constexpr size_t smem = 65536 + 1024;   // (k1_lane_launch: histogram + tail / output row / class tables of the flush)
    static lrb_per_device_once attr_many;
    if (attr_many.need(c->device))
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k1_lane4s2_kernel<8, 2, false, 4, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL((k1_lane4s2_kernel<8, 2, false, 4, true>), dim3((unsigned)(2 * groups)), dim3(512), smem, c->stream,
                       reinterpret_cast<const uint4 *>(base), (const uint64_t *)d_pairs, (const uint32_t *)d_order,
                       (const uint32_t *)d_lens, groups * 64, d_counts, (uint64_t *)nullptr);
    HIP_TRY(hipGetLastError());
    return LRB_OK;
}

// Many resident batches into the table, forward tallies: one direct-kernel launch a batch (see lrb_k15_accumulate_part_dev)
extern "C" int lrb_packed_k15_accumulate_many(lrb_ctx *c, const lrb_packed *const *ps, uint64_t count, uint32_t *d_table)
{
    ARG_TRY(c != nullptr && d_table != nullptr && (ps != nullptr || count == 0));
    HIP_TRY(hipSetDevice(c->device));
    for (uint64_t i = 0; i < count; ++i) {
        const lrb_packed *p = ps[i];
        ARG_TRY(p != nullptr);
        if (p->n == 0 || p->total_bases == 0) continue;
        const int rc = lrb_k15_accumulate_dev(c, p->pd.codes, p->pd.mask, p->pd.code_off, p->pd.mask_off, p->pd.lens, p->n, d_table);
        if (rc != LRB_OK) return rc;
    }
    return LRB_OK;
}

extern "C" int lrb_packed_k15_accumulate(lrb_ctx *c, const lrb_packed *p, uint32_t *d_table)
{
    ARG_TRY(c != nullptr && p != nullptr && d_table != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    if (p->n == 0) return LRB_OK;
    return lrb_k15_accumulate_dev(c, p->pd.codes, p->pd.mask, p->pd.code_off, p->pd.mask_off, p->pd.lens, p->n, d_table);
}

extern "C" int lrb_packed_cov_hist(lrb_ctx *c, const lrb_packed *p, const uint32_t *d_table,
                                   int64_t bin_size, int bins, uint32_t *hist, uint32_t *sums)
{
    ARG_TRY(c != nullptr && p != nullptr && d_table != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    ARG_TRY(bin_size >= 1);
    ARG_TRY(bins >= 1 && bins <= 1024);
    if (p->n == 0) return LRB_OK;
    ARG_TRY(hist && sums);
    void *d_hist, *d_sums;
    int rc = ws_get(c, 5, sizeof(uint32_t) * p->n * bins, &d_hist);
    if (rc != LRB_OK) return rc;
    rc = ws_get(c, 6, sizeof(uint32_t) * p->n, &d_sums);
    if (rc != LRB_OK) return rc;
    rc = lrb_cov_hist_dev(c, p->pd.codes, p->pd.mask, p->pd.code_off, p->pd.mask_off, p->pd.lens,
                          p->n, d_table, bin_size, bins, (uint32_t *)d_hist, (uint32_t *)d_sums);
    if (rc != LRB_OK) return rc;
    rc = lrb_copy_d2h(c, hist, d_hist, sizeof(uint32_t) * p->n * bins);
    if (rc != LRB_OK) return rc;
    return lrb_copy_d2h(c, sums, d_sums, sizeof(uint32_t) * p->n);
}

// K3 of MANY resident batches as one sweep.  A batch of the reader is a few thousand reads -- too few for the
// sweep to fill the CUs with read groups -- so the batches' codes, masks and lengths are laid end to end in
// workspace (device copies: 375 bytes per 1000 bases) with their offsets rebased, and lrb_cov_hist_sweep_dev runs
// on the lot.  The histograms stay in the context (slots 5 / 6, rows in batch order) for lrb_cov_rows_text.
// masks and lengths of `count` batches laid end to end, offsets rebased: ONE launch (blockIdx.y = batch) instead of
// three copies and a kernel per batch -- 240 enqueues per group of sixty batches were 7-10 ms of a C4 rank's table and
// coverage phases each
__global__ __launch_bounds__(256) void concat_packs_kernel(const pack_desc *__restrict__ descs, uint32_t *__restrict__ mask_out,
                                                           uint32_t *__restrict__ lens_out, uint64_t *__restrict__ code_out,
                                                           uint64_t *__restrict__ mask_off_out)
{
    const pack_desc d = descs[blockIdx.y];
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    const uint4 *m4 = reinterpret_cast<const uint4 *>(d.mask);      // (mask regions are multiples of four words)
    uint4 *o4 = reinterpret_cast<uint4 *>(mask_out + d.mask0);
    const uint64_t n4 = d.mask_words / 4;
    uint64_t i = t;
    for (; i + 3 * stride < n4; i += 4 * stride) {   // four independent 16-byte loads in flight per thread
        const uint4 a = m4[i], b = m4[i + stride], c_ = m4[i + 2 * stride], e = m4[i + 3 * stride];
        o4[i] = a;
        o4[i + stride] = b;
        o4[i + 2 * stride] = c_;
        o4[i + 3 * stride] = e;
    }
    for (; i < n4; i += stride) o4[i] = m4[i];
    for (uint64_t j = (d.mask_words & ~3ull) + t; j < d.mask_words; j += stride) mask_out[d.mask0 + j] = d.mask[j];
    for (uint64_t j = t; j < d.n; j += stride) lens_out[d.read0 + j] = d.lens[j];
    // entries 0..n-1, and entry n (the end) for the last batch only: the next batch's entry 0 is the same value
    const uint64_t m = d.n + (d.last ? 1 : 0);
    for (uint64_t j = t; j < m; j += stride) {
        code_out[d.read0 + j] = d.code_off[j] + d.code_delta;
        mask_off_out[d.read0 + j] = d.mask_off[j] + d.mask0;
    }
}

// (diagnosis, round 6: the per-batch form of round 5 -- LRB_CONCAT_KERNEL=0)
__global__ __launch_bounds__(256) void rebase_offsets_kernel(const uint64_t *__restrict__ code_off,
                                                             const uint64_t *__restrict__ mask_off, uint64_t n,
                                                             uint64_t code_base, uint64_t mask_base,
                                                             uint64_t *__restrict__ code_out,
                                                             uint64_t *__restrict__ mask_out, int last)
{
    // entries 0..n-1, and entry n (the end) for the last batch only: the next batch's entry 0 is the same value
    const uint64_t m = n + (last ? 1 : 0);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (uint64_t)gridDim.x * blockDim.x) {
        code_out[i] = code_off[i] + code_base;
        mask_out[i] = mask_off[i] + mask_base;
    }
}

// The packed reads of `count` batches as ONE batch: masks and lengths copied end to end (the list kernels walk a unit's
// mask words as one run), offsets rebased.  The CODES -- eight ninths of the bytes -- are copied only when d_codes is
// given: every kernel reaches a read's codes as base + code_off[read], so with d_codes == nullptr the base is the
// lowest of the batches' own buffers and a read's offset the distance from there (*codes_base; the batches must outlive
// what is made from it).  The table of batches goes up through workspace slot 0.
static int concat_packs(lrb_ctx *c, const lrb_packed *const *packs, uint64_t count, uint32_t *d_codes, uint32_t *d_mask,
                        uint64_t *d_co, uint64_t *d_mo, uint32_t *d_lens, const uint32_t **codes_base)
{
    uint64_t at = 0, cb = 0, mb = 0, last = count;
    const uint32_t *base = d_codes;
    for (uint64_t i = 0; i < count; ++i)
        if (packs[i]->n) {
            last = i;
            if (!d_codes && (!base || packs[i]->pd.codes < base)) base = packs[i]->pd.codes;
        }
    static const bool one_kernel = !(getenv("LRB_CONCAT_KERNEL") && atoi(getenv("LRB_CONCAT_KERNEL")) == 0);
    if (!one_kernel) {
        for (uint64_t i = 0; i < count; ++i) {
            const lrb_packed *p = packs[i];
            if (p->n == 0) continue;
            if (d_codes)
                HIP_TRY(hipMemcpyAsync(d_codes + cb, p->pd.codes, sizeof(uint32_t) * p->code_words, hipMemcpyDeviceToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(d_mask + mb, p->pd.mask, sizeof(uint32_t) * p->mask_words, hipMemcpyDeviceToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(d_lens + at, p->pd.lens, sizeof(uint32_t) * p->n, hipMemcpyDeviceToDevice, c->stream));
            unsigned blocks = (unsigned)((p->n + 256) / 256);
            if (blocks > 1024) blocks = 1024;
            hipLaunchKernelGGL(rebase_offsets_kernel, dim3(blocks), dim3(256), 0, c->stream, p->pd.code_off, p->pd.mask_off,
                               p->n, d_codes ? cb : (uint64_t)(p->pd.codes - base), mb, d_co + at, d_mo + at, i == last ? 1 : 0);
            at += p->n;
            cb += p->code_words;
            mb += p->mask_words;
        }
        HIP_TRY(hipGetLastError());
        if (codes_base) *codes_base = base;
        return LRB_OK;
    }
    std::vector<pack_desc> descs;
    descs.reserve(count);
    for (uint64_t i = 0; i < count; ++i) {
        const lrb_packed *p = packs[i];
        if (p->n == 0) continue;
        if (d_codes)
            HIP_TRY(hipMemcpyAsync(d_codes + cb, p->pd.codes, sizeof(uint32_t) * p->code_words, hipMemcpyDeviceToDevice, c->stream));
        pack_desc d = {};
        d.mask = p->pd.mask;
        d.lens = p->pd.lens;
        d.code_off = p->pd.code_off;
        d.mask_off = p->pd.mask_off;
        d.n = p->n;
        d.mask_words = p->mask_words;
        d.read0 = at;
        d.mask0 = mb;
        d.code_delta = d_codes ? cb : (uint64_t)(p->pd.codes - base);
        d.last = i == last ? 1u : 0u;
        descs.push_back(d);
        at += p->n;
        cb += p->code_words;
        mb += p->mask_words;
    }
    if (!descs.empty()) {
        void *d_descs;
        const int rc = lrb_stage_upload(c, 0, descs.data(), sizeof(pack_desc) * descs.size(), &d_descs);
        if (rc != LRB_OK) return rc;
        // (a batch of the parser pool is ~6,700 reads and ~2 M mask words: 32 blocks of 256 threads take eight passes)
        // (A/B of round 6: LRB_CONCAT_BLOCKS = workgroups per batch, default 64)
        static const unsigned bpb = getenv("LRB_CONCAT_BLOCKS") ? (unsigned)atoi(getenv("LRB_CONCAT_BLOCKS")) : 64u;
        for (size_t d0 = 0; d0 < descs.size(); d0 += 65535) {
            const size_t nd = descs.size() - d0 < 65535 ? descs.size() - d0 : 65535;
            hipLaunchKernelGGL(concat_packs_kernel, dim3(bpb ? bpb : 64u, (unsigned)nd), dim3(256), 0, c->stream,
                               (const pack_desc *)d_descs + d0, d_mask, d_lens, d_co, d_mo);
        }
        HIP_TRY(hipGetLastError());
    }
    if (codes_base) *codes_base = base;
    return LRB_OK;
}

static bool resident_lists_match(const lrb_ctx *c, const lrb_packed *const *packs, uint64_t count, int bins);

extern "C" int lrb_packed_cov_hist_many(lrb_ctx *c, const lrb_packed *const *packs, uint64_t count, const uint8_t *d_map,
                                        int bins)
{
    ARG_TRY(c != nullptr && d_map != nullptr && (count == 0 || packs != nullptr));
    HIP_TRY(hipSetDevice(c->device));
    ARG_TRY(bins >= 1 && bins <= 256);
    uint64_t n = 0, mw = 0;
    for (uint64_t i = 0; i < count; ++i) {
        ARG_TRY(packs[i] != nullptr);
        n += packs[i]->n;
        mw += packs[i]->n ? packs[i]->mask_words : 0;
    }
    if (n == 0) return LRB_OK;
    if (resident_lists_match(c, packs, count, bins)) return lrb_winlists_cov_hist(c, c->res_lists, d_map, bins);
    void *d_mask, *d_offs, *d_lens, *d_hist, *d_sums;
    int rc = ws_get(c, 13, sizeof(uint32_t) * (mw + 16), &d_mask);
    if (rc == LRB_OK) rc = ws_get(c, 14, sizeof(uint64_t) * (n + 1) * 2, &d_offs);
    if (rc == LRB_OK) rc = ws_get(c, 15, sizeof(uint32_t) * n, &d_lens);
    if (rc == LRB_OK) rc = ws_get(c, 5, sizeof(uint32_t) * n * bins, &d_hist);
    if (rc == LRB_OK) rc = ws_get(c, 6, sizeof(uint32_t) * n, &d_sums);
    if (rc != LRB_OK) return rc;
    uint64_t *d_co = (uint64_t *)d_offs, *d_mo = d_co + (n + 1);
    const uint32_t *d_codes = nullptr; // (the batches' own codes, reached from the lowest of their buffers)
    rc = concat_packs(c, packs, count, nullptr, (uint32_t *)d_mask, d_co, d_mo, (uint32_t *)d_lens, &d_codes);
    if (rc != LRB_OK) return rc;
    return lrb_cov_hist_sweep_dev(c, d_codes, (const uint32_t *)d_mask, d_co, d_mo,
                                  (const uint32_t *)d_lens, n, d_map, bins, (uint32_t *)d_hist, (uint32_t *)d_sums);
}

// ---- the windows of MANY resident batches partitioned once, for K2's tally and K3's sweep (round 3) ----
struct lrb_winlists {
    void *mem[7]; // codes, mask, offsets (code | mask), lens, lists, bounds, group bases (all null: in the workspace)
    uint64_t mem_size[7];
    uint64_t epoch; // in the workspace: the context's lists_epoch when they were made
    bool in_ws;
    uint32_t *codes, *mask, *lens, *lists, *bounds;
    uint64_t *code_off, *mask_off, *gbase;
    uint64_t n, bytes, total_bases, ngroups;
    uint32_t R;
    int device;
};

extern "C" int lrb_winlists_free(lrb_ctx *c, lrb_winlists *w)
{
    if (!w) return LRB_OK;
    if (c) (void)hipSetDevice(c->device);
    if (c && c->pool_cap) (void)hipStreamSynchronize(c->stream); // (a retained block is handed out again without hipFree's wait)
    for (int i = 0; i < 7; ++i) pool_give(c, w->mem[i], w->mem_size[i]);
    delete w;
    return LRB_OK;
}

static bool winlists_stale(const lrb_ctx *c, const lrb_winlists *w) { return w->in_ws && w->epoch != c->lists_epoch; }

extern "C" int lrb_packed_lists_create(lrb_ctx *c, const lrb_packed *const *packs, uint64_t count, int bins,
                                       int in_workspace, lrb_winlists **out)
{
    ARG_TRY(c != nullptr && out != nullptr && (count == 0 || packs != nullptr));
    HIP_TRY(hipSetDevice(c->device));
    ARG_TRY(bins >= 1 && bins <= 256);
    *out = nullptr;
    uint64_t n = 0, cw = 0, mw = 0, bases = 0;
    for (uint64_t i = 0; i < count; ++i) {
        ARG_TRY(packs[i] != nullptr);
        n += packs[i]->n;
        cw += packs[i]->n ? packs[i]->code_words : 0;
        mw += packs[i]->n ? packs[i]->mask_words : 0;
        bases += packs[i]->total_bases;
    }
    ARG_TRY(bases <= 0xFFFFFFFFull); // a bucket's uint32 tally
    lrb_winlists *w = new (std::nothrow) lrb_winlists();
    if (!w) return LRB_ERR_NOMEM;
    w->n = n;
    w->total_bases = bases;
    w->device = c->device;
    w->R = (uint32_t)lrb_wl_group_reads(c, n ? n : 1, bins, bases);
    w->ngroups = (n + w->R - 1) / w->R;
    const uint64_t sizes[7] = {sizeof(uint32_t) * (cw + 16), sizeof(uint32_t) * (mw + 16), sizeof(uint64_t) * (n + 1) * 2,
                               sizeof(uint32_t) * (n + 1), sizeof(uint32_t) * (mw * 32 + 16),
                               sizeof(uint32_t) * (lrb_k15_lists_bounds_words(w->ngroups) + 4),
                               sizeof(uint64_t) * (w->ngroups + 2)};
    void *at[7] = {};
    if (in_workspace) {
        // in the context's workspaces (no allocation once they have grown): valid until the next call that uses them
        // (the codes stay where the batches hold them: a transient list's kernels reach them from there)
        const int slot[5] = {12, 13, 14, 15, 8};
        for (int i = 1; i < 5; ++i) {
            const int rc = ws_get(c, slot[i], sizes[i], &at[i]);
            if (rc != LRB_OK) {
                delete w;
                return rc;
            }
        }
        void *small;
        const int rc = ws_get(c, 11, sizes[5] + sizes[6] + 64, &small);
        if (rc != LRB_OK) {
            delete w;
            return rc;
        }
        at[6] = small;
        at[5] = (char *)small + ((sizes[6] + 15) & ~15ull);
        w->in_ws = true;
        w->epoch = c->lists_epoch;
    } else {
        for (int i = 0; i < 7; ++i) {
            if (pool_take(c, sizes[i], &w->mem[i], &w->mem_size[i]) != hipSuccess) {
                w->mem[i] = nullptr;
                (void)hipGetLastError();
                lrb_winlists_free(c, w);
                lrb_set_error("slice lists: out of device memory%s%s", "", "");
                return LRB_ERR_NOMEM;
            }
            at[i] = w->mem[i];
        }
    }
    for (int i = in_workspace ? 1 : 0; i < 7; ++i) w->bytes += sizes[i];
    w->codes = (uint32_t *)at[0];
    w->mask = (uint32_t *)at[1];
    w->code_off = (uint64_t *)at[2];
    w->mask_off = w->code_off + (n + 1);
    w->lens = (uint32_t *)at[3];
    w->lists = (uint32_t *)at[4];
    w->bounds = (uint32_t *)at[5];
    w->gbase = (uint64_t *)at[6];
    int rc = LRB_OK;
    if (n) {
        const uint32_t *base = nullptr;
        rc = concat_packs(c, packs, count, w->codes, w->mask, w->code_off, w->mask_off, w->lens, &base);
        w->codes = const_cast<uint32_t *>(base);
        if (rc == LRB_OK)
            rc = lrb_k15_lists_part_dev(c, w->codes, w->mask, w->code_off, w->mask_off, w->lens, n, w->R, w->lists, w->bounds,
                                        w->gbase);
    }
    if (rc != LRB_OK) {
        lrb_winlists_free(c, w);
        return rc;
    }
    *out = w;
    return LRB_OK;
}

extern "C" int lrb_winlists_valid(const lrb_ctx *c, const lrb_winlists *w, int *valid)
{
    ARG_TRY(c != nullptr && w != nullptr && valid != nullptr);
    *valid = winlists_stale(c, w) ? 0 : 1;
    return LRB_OK;
}

extern "C" int lrb_winlists_info(const lrb_winlists *w, uint64_t *n_reads, uint64_t *device_bytes, uint32_t *reads_per_group)
{
    ARG_TRY(w != nullptr);
    if (n_reads) *n_reads = w->n;
    if (device_bytes) *device_bytes = w->bytes;
    if (reads_per_group) *reads_per_group = w->R;
    return LRB_OK;
}

extern "C" int lrb_winlists_tally(lrb_ctx *c, const lrb_winlists *w, uint32_t *d_half)
{
    ARG_TRY(c != nullptr && w != nullptr && d_half != nullptr && w->device == c->device);
    if (winlists_stale(c, w)) {
        lrb_set_error("slice lists: the workspace they were made in has been used since%s%s", "", "");
        return LRB_ERR_ARG;
    }
    if (w->n == 0) return LRB_OK;
    return lrb_k15_lists_tally_dev(c, w->codes, w->mask, w->code_off, w->mask_off, w->lens, w->n, w->R, w->lists, w->bounds,
                                   w->gbase, d_half);
}

/* K3 of the partitioned reads: histograms into the context (slots 5 / 6, rows in batch order) for lrb_cov_rows_text */
extern "C" int lrb_winlists_cov_hist(lrb_ctx *c, const lrb_winlists *w, const uint8_t *d_map, int bins)
{
    ARG_TRY(c != nullptr && w != nullptr && d_map != nullptr && w->device == c->device);
    HIP_TRY(hipSetDevice(c->device));
    ARG_TRY(lrb_wl_hist_fits(w->R, bins));
    if (winlists_stale(c, w)) {
        lrb_set_error("slice lists: the workspace they were made in has been used since%s%s", "", "");
        return LRB_ERR_ARG;
    }
    if (w->n == 0) return LRB_OK;
    void *d_hist, *d_sums;
    int rc = ws_get(c, 5, sizeof(uint32_t) * w->n * bins, &d_hist);
    if (rc == LRB_OK) rc = ws_get(c, 6, sizeof(uint32_t) * w->n, &d_sums);
    if (rc != LRB_OK) return rc;
    return lrb_cov_lists_sweep_dev(c, w->codes, w->mask, w->code_off, w->mask_off, w->lens, w->n, w->R, w->lists, w->bounds,
                                   w->gbase, d_map, bins, (uint32_t *)d_hist, (uint32_t *)d_sums);
}

// ---- the last group's lists stay in the workspace for the coverage stage (round 6) ----
// The record of the lists kept in the workspace is touched from more than one thread: the runners give resident batches
// back on a thread of their own (release_resident(background=True): lrb_packed_free per batch) while the main thread trims
// the context -- both forget the lists.  One mutex for every context's record (the operations are a few pointer moves).
static std::mutex g_resident_mu;

static void resident_lists_drop_locked(lrb_ctx *c)
{
    lrb_winlists *w = c->res_lists;
    c->res_lists = nullptr;
    free((void *)c->res_packs);
    c->res_packs = nullptr;
    c->res_count = 0;
    delete w;   // (made in the workspace: the object owns no device memory, nothing to give back to the pool)
}

void lrb_resident_lists_drop(lrb_ctx *c)
{
    if (!c) return;
    std::lock_guard<std::mutex> lk(g_resident_mu);
    resident_lists_drop_locked(c);
}

void lrb_resident_lists_forget_batch(lrb_ctx *c, const lrb_packed *p)
{
    std::lock_guard<std::mutex> lk(g_resident_mu);
    for (uint64_t i = 0; i < c->res_count; ++i)
        if (c->res_packs[i] == p) {
            resident_lists_drop_locked(c);
            return;
        }
}

static bool resident_lists_on()
{
    const char *e = getenv("LRB_RESIDENT_LISTS");
    return !(e && atoi(e) == 0);
}

// the lists kept in the workspace were made from exactly these batches, still stand, and hold a histogram of `bins`
static bool resident_lists_match(const lrb_ctx *c, const lrb_packed *const *packs, uint64_t count, int bins)
{
    std::lock_guard<std::mutex> lk(g_resident_mu);
    if (!c->res_lists || !resident_lists_on() || winlists_stale(c, c->res_lists)) return false;
    uint64_t j = 0;
    for (uint64_t i = 0; i < count; ++i) {   // (empty batches are in neither)
        if (!packs[i] || packs[i]->n == 0) continue;
        if (j >= c->res_count || c->res_packs[j] != packs[i]) return false;
        ++j;
    }
    return j == c->res_count && lrb_wl_hist_fits(c->res_lists->R, bins);
}

extern "C" int lrb_packed_lists_resident(const lrb_ctx *c, const lrb_packed *const *packs, uint64_t count, int bins, int *yes)
{
    ARG_TRY(c != nullptr && yes != nullptr && (packs != nullptr || count == 0));
    *yes = count && resident_lists_match(c, packs, count, bins) ? 1 : 0;
    return LRB_OK;
}

// Where lrb_packed_k15_tally_half_many cuts `count` consecutive batches into groups of at most max_bases bases: filled
// FROM THE END, so that the last group -- the one whose lists stay in the workspace -- is a full one (432,333 reads of
// 10 kb: 32 k + 400 k, not 400 k + 32 k).  starts[g] = first batch of group g, starts[*n_groups] = count; the caller
// gives count + 1 entries.
extern "C" int lrb_packed_group_starts(const lrb_packed *const *packs, uint64_t count, uint64_t max_bases, uint64_t *starts,
                                       uint64_t *n_groups)
{
    ARG_TRY(starts != nullptr && n_groups != nullptr && (packs != nullptr || count == 0));
    uint64_t ng = 0, g1 = count;
    while (g1 > 0) {   // group [g0, g1): grown downwards from g1
        uint64_t g0 = g1, bases = 0;
        while (g0 > 0) {
            ARG_TRY(packs[g0 - 1] != nullptr);
            if (g0 < g1 && bases + packs[g0 - 1]->total_bases > max_bases) break;
            bases += packs[g0 - 1]->total_bases;
            --g0;
        }
        starts[ng++] = g0;    // (descending for now)
        g1 = g0;
    }
    for (uint64_t i = 0; i < ng / 2; ++i) {
        const uint64_t t = starts[i];
        starts[i] = starts[ng - 1 - i];
        starts[ng - 1 - i] = t;
    }
    starts[ng] = count;
    *n_groups = ng;
    return LRB_OK;
}

// K2 of MANY resident batches as the product runs it: consecutive batches in groups of at most 4e9 bases (16 GB of
// lists in the context's workspaces; lrb_packed_group_starts), each group partitioned once and tallied into the canonical
// half; a group of fewer than LRB_K2_LISTS_MIN_BASES bases (default 33 M) by one atomic a window.  What
// runners_utils.run_15mer_counts, the sharded driver and the count-15mers executable call when they do not keep the
// lists.  `bins`: the histogram the coverage stage will want (lists made for it hold any narrower one); the LAST group's
// lists are left standing in the workspace for lrb_packed_cov_hist_many of the same batches (LRB_RESIDENT_LISTS=0: not).
extern "C" int lrb_packed_k15_tally_half_many_for(lrb_ctx *c, const lrb_packed *const *packs, uint64_t count, uint32_t *d_half,
                                                  int bins)
{
    ARG_TRY(c != nullptr && d_half != nullptr && (packs != nullptr || count == 0));
    HIP_TRY(hipSetDevice(c->device));
    ARG_TRY(bins >= 1 && bins <= 256);
    lrb_resident_lists_drop(c);
    if (count == 0) return LRB_OK;
    uint64_t min_bases = 33000000ull;
    if (const char *e = getenv("LRB_K2_LISTS_MIN_BASES")) min_bases = strtoull(e, nullptr, 10);
    uint64_t group_bases = 4000000000ull;
    if (const char *e = getenv("LRB_K2_GROUP_BASES")) group_bases = strtoull(e, nullptr, 10);   // (tests: several groups of a small file)
    if (group_bases < 1 || group_bases > 0xFFFFFFFFull) group_bases = 4000000000ull;
    uint64_t *starts = (uint64_t *)malloc(sizeof(uint64_t) * (count + 1));
    if (!starts) return LRB_ERR_NOMEM;
    uint64_t ng = 0;
    int rc = lrb_packed_group_starts(packs, count, group_bases, starts, &ng);
    if (rc == LRB_OK && ng > 1) {
        // the workspaces at the size of the LARGEST group before the first (remainder) group takes them: lists, laid-out
        // masks / offsets / lengths and the level-1 scratch would otherwise be allocated small, then again large, and the
        // scratch -- kept at its first size -- would cut every full group into chunks
        uint64_t n_max = 0, mw_max = 0;
        for (uint64_t g = 0; g < ng; ++g) {
            uint64_t n_g = 0, mw_g = 0, bases = 0;
            for (uint64_t i = starts[g]; i < starts[g + 1]; ++i) {
                n_g += packs[i]->n;
                mw_g += packs[i]->n ? packs[i]->mask_words : 0;
                bases += packs[i]->total_bases;
            }
            if (bases >= min_bases && bases <= 0xFFFFFFFFull && mw_g > mw_max) n_max = n_g, mw_max = mw_g;
        }
        if (mw_max) {
            void *p;
            const int slot[4] = {13, 14, 15, 8};
            const uint64_t sizes[4] = {sizeof(uint32_t) * (mw_max + 16), sizeof(uint64_t) * (n_max + 1) * 2, sizeof(uint32_t) * (n_max + 1),
                                       sizeof(uint32_t) * (mw_max * 32 + 16)};
            for (int i = 0; i < 4 && rc == LRB_OK; ++i) rc = ws_get(c, slot[i], sizes[i], &p);
            if (rc == LRB_OK) rc = lrb_wl_reserve_scratch(c, mw_max * 32);
        }
    }
    for (uint64_t g = 0; g < ng && rc == LRB_OK; ++g) {
        const uint64_t g0 = starts[g], g1 = starts[g + 1];
        uint64_t bases = 0;
        for (uint64_t i = g0; i < g1; ++i) bases += packs[i]->total_bases;
        if (bases < min_bases || bases > 0xFFFFFFFFull) {
            for (uint64_t i = g0; i < g1 && rc == LRB_OK; ++i) rc = lrb_packed_k15_accumulate_half(c, packs[i], d_half);
            continue;
        }
        lrb_winlists *w = nullptr;
        rc = lrb_packed_lists_create(c, packs + g0, g1 - g0, bins < 32 ? 32 : bins, 1, &w);
        if (rc == LRB_OK) rc = lrb_winlists_tally(c, w, d_half);
        if (rc == LRB_OK && g + 1 == ng && resident_lists_on()) {
            // the last group: its lists, bounds and laid-out masks stay where they are
            const lrb_packed **keep = (const lrb_packed **)malloc(sizeof(lrb_packed *) * (g1 - g0));
            if (keep) {
                uint64_t j = 0;
                for (uint64_t i = g0; i < g1; ++i)
                    if (packs[i]->n) keep[j++] = packs[i];
                std::lock_guard<std::mutex> lk(g_resident_mu);
                c->res_lists = w;
                c->res_packs = keep;
                c->res_count = j;
                w = nullptr;
            }
        }
        (void)lrb_winlists_free(c, w);
    }
    free(starts);
    return rc;
}

extern "C" int lrb_packed_k15_tally_half_many(lrb_ctx *c, const lrb_packed *const *packs, uint64_t count, uint32_t *d_half)
{
    return lrb_packed_k15_tally_half_many_for(c, packs, count, d_half, 32);
}

extern "C" int lrb_packed_k15_accumulate_half(lrb_ctx *c, const lrb_packed *p, uint32_t *d_half)
{
    ARG_TRY(c != nullptr && p != nullptr && d_half != nullptr);
    if (p->n == 0) return LRB_OK;
    return lrb_k15_accumulate_half_dev(c, p->pd.codes, p->pd.mask, p->pd.code_off, p->pd.mask_off, p->pd.lens, p->n, d_half);
}

static int packed_text_out(lrb_ctx *c, int mode, const uint32_t *d_vals, const uint32_t *d_per_row, uint64_t n, uint32_t dim, int k,
                           uint8_t *text, uint32_t *q6);

/* cov_profs rows [first_row, first_row + n_rows) of the histograms lrb_packed_cov_hist_many left in the context. */
extern "C" int lrb_cov_rows_text(lrb_ctx *c, uint64_t first_row, uint64_t n_rows, int bins, uint8_t *text, uint32_t *q6)
{
    ARG_TRY(c != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    ARG_TRY(bins >= 1 && bins <= 256);
    if (n_rows == 0) return LRB_OK;
    ARG_TRY(text != nullptr);
    ARG_TRY(c->ws[5] && c->ws[6] && c->ws_bytes[5] >= sizeof(uint32_t) * (first_row + n_rows) * bins &&
            c->ws_bytes[6] >= sizeof(uint32_t) * (first_row + n_rows));
    return packed_text_out(c, 1, (const uint32_t *)c->ws[5] + first_row * bins, (const uint32_t *)c->ws[6] + first_row,
                           n_rows, (uint32_t)bins, 0, text, q6);
}

// ---- the same stages ending in text (K8, lrb_format.hip) ---------------------
static int packed_text_out(lrb_ctx *c, int mode, const uint32_t *d_vals, const uint32_t *d_per_row, uint64_t n, uint32_t dim, int k,
                           uint8_t *text, uint32_t *q6)
{
    const uint64_t width = mode == 0 ? lrb_com_row_bytes(dim) : lrb_cov_row_bytes(dim);
    void *d_text, *d_q = nullptr;
    int rc = ws_get(c, 7, n * width + 16, &d_text);
    if (rc != LRB_OK) return rc;
    if (q6) {
        rc = ws_get(c, 4, sizeof(uint32_t) * n * dim, &d_q);
        if (rc != LRB_OK) return rc;
    }
    rc = mode == 0 ? lrb_format_com_dev(c, d_vals, d_per_row, n, dim, k, (uint8_t *)d_text, (uint32_t *)d_q)
                   : lrb_format_cov_dev(c, d_vals, d_per_row, n, dim, (uint8_t *)d_text, (uint32_t *)d_q);
    if (rc != LRB_OK) return rc;
    rc = lrb_copy_d2h(c, text, d_text, n * width);
    if (rc != LRB_OK || !q6) return rc;
    return lrb_copy_d2h(c, q6, d_q, sizeof(uint32_t) * n * dim);
}

// The same two text stages for a host batch that is NOT kept (the three executables of the reference's
// process boundary parse the file once each): upload, pack, tally and format in the context's workspaces,
// no allocation per batch.
extern "C" int lrb_kmer_text_host(lrb_ctx *c, const uint8_t *seqs, const uint64_t *offs, uint64_t n, int k,
                                  uint8_t *text, uint32_t *q6)
{
    ARG_TRY(c != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    ARG_TRY(k >= 3 && k <= 5);
    if (n == 0) return LRB_OK;
    ARG_TRY(seqs && offs && text);
    packed_dev pd;
    int rc = upload_and_pack(c, seqs, offs, n, false, k == 3 ? 1 : 2, &pd);
    if (rc != LRB_OK) return rc;
    void *d_counts;
    rc = ws_get(c, 5, sizeof(uint32_t) * n * c->dim[k], &d_counts);
    if (rc != LRB_OK) return rc;
    if (k == 3)
        rc = lrb_kmer_counts3t_dev(c, pd.planes_t, pd.group_off, pd.order, pd.lens, n, (uint32_t *)d_counts);
    else
        rc = lrb_kmer_counts_t_dev(c, k, pd.codes_t, pd.group_off4, pd.order4, pd.lens, n, (uint32_t *)d_counts);
    if (rc != LRB_OK) return rc;
    // the text goes to slot 7 and q6 to slot 4 -- the planes / their order of k = 3, which the tally (ahead of the
    // formatter on the stream) has finished with
    return packed_text_out(c, 0, (const uint32_t *)d_counts, pd.lens, n, c->dim[k], k, text, q6);
}

extern "C" int lrb_cov_text_host(lrb_ctx *c, const uint8_t *seqs, const uint64_t *offs, uint64_t n,
                                 const uint32_t *d_table, int64_t bin_size, int bins, uint8_t *text, uint32_t *q6)
{
    ARG_TRY(c != nullptr && d_table != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    ARG_TRY(bin_size >= 1);
    ARG_TRY(bins >= 1 && bins <= 1024);
    if (n == 0) return LRB_OK;
    ARG_TRY(seqs && offs && text);
    packed_dev pd;
    int rc = upload_and_pack(c, seqs, offs, n, true, 0, &pd);
    if (rc != LRB_OK) return rc;
    void *d_hist, *d_sums;
    rc = ws_get(c, 5, sizeof(uint32_t) * n * bins, &d_hist);
    if (rc != LRB_OK) return rc;
    rc = ws_get(c, 6, sizeof(uint32_t) * n, &d_sums);
    if (rc != LRB_OK) return rc;
    rc = lrb_cov_hist_dev(c, pd.codes, pd.mask, pd.code_off, pd.mask_off, pd.lens, n, d_table, bin_size, bins,
                          (uint32_t *)d_hist, (uint32_t *)d_sums);
    if (rc != LRB_OK) return rc;
    // q6 would go to slot 4, the mask the coverage kernel has finished with by then
    return packed_text_out(c, 1, (const uint32_t *)d_hist, (const uint32_t *)d_sums, n, (uint32_t)bins, 0, text, q6);
}

extern "C" int lrb_packed_kmer_text(lrb_ctx *c, const lrb_packed *p, int k, uint8_t *text, uint32_t *q6)
{
    ARG_TRY(c != nullptr && p != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    ARG_TRY(k >= 3 && k <= 5);
    if (p->n == 0) return LRB_OK;
    ARG_TRY(text != nullptr);
    void *d_counts;
    int rc = ws_get(c, 5, sizeof(uint32_t) * p->n * c->dim[k], &d_counts);
    if (rc != LRB_OK) return rc;
    if (k == 3 && p->has_planes)
        rc = lrb_kmer_counts3t_dev(c, p->pd.planes_t, p->pd.group_off, p->pd.order, p->pd.lens, p->n,
                                   (uint32_t *)d_counts);
    else if (k != 3 && p->has_codes_t)
        rc = lrb_kmer_counts_t_dev(c, k, p->pd.codes_t, p->pd.group_off4, p->pd.order4, p->pd.lens, p->n,
                                   (uint32_t *)d_counts);
    else
        rc = lrb_kmer_counts_dev(c, p->pd.codes, p->pd.code_off, p->pd.lens, p->n, k,
                                 (uint32_t *)d_counts);
    if (rc != LRB_OK) return rc;
    return packed_text_out(c, 0, (const uint32_t *)d_counts, p->pd.lens, p->n, c->dim[k], k, text, q6);
}

extern "C" int lrb_packed_cov_text(lrb_ctx *c, const lrb_packed *p, const uint32_t *d_table,
                                   int64_t bin_size, int bins, uint8_t *text, uint32_t *q6)
{
    ARG_TRY(c != nullptr && p != nullptr && d_table != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    ARG_TRY(bin_size >= 1);
    ARG_TRY(bins >= 1 && bins <= 1024);
    if (p->n == 0) return LRB_OK;
    ARG_TRY(text != nullptr);
    void *d_hist, *d_sums;
    int rc = ws_get(c, 5, sizeof(uint32_t) * p->n * bins, &d_hist);
    if (rc != LRB_OK) return rc;
    rc = ws_get(c, 6, sizeof(uint32_t) * p->n, &d_sums);
    if (rc != LRB_OK) return rc;
    rc = lrb_cov_hist_dev(c, p->pd.codes, p->pd.mask, p->pd.code_off, p->pd.mask_off, p->pd.lens,
                          p->n, d_table, bin_size, bins, (uint32_t *)d_hist, (uint32_t *)d_sums);
    if (rc != LRB_OK) return rc;
    return packed_text_out(c, 1, (const uint32_t *)d_hist, (const uint32_t *)d_sums, p->n, (uint32_t)bins, 0, text, q6);
}
