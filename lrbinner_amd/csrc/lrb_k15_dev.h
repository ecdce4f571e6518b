// lrb_k15_dev.h -- device helpers shared by the HIP translation units that walk 15-mer windows
// (lrb_kernels.hip: K1 / direct K2 / gather K3; lrb_lists.hip: K2 + K3 on one partition of the windows).
#ifndef LRB_K15_DEV_H
#define LRB_K15_DEV_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define WAVE 64
#define K15_MASK 0x3FFFFFFFu

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & (WAVE - 1); }

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

// ---------------------------------------------------------------------------
// 15-mer window helpers shared by K2 and K3.  One lane owns a 32-base chunk.
// ---------------------------------------------------------------------------
// bit (31-i) of the result: the 15-mer starting at base i of the chunk is valid, i.e.
// mask bits i..i+14 are all one (run-length test by doubling: 2,4,8,15) -- the reset rule of
// kmer_utils.h:38-43,122-127.
__device__ __forceinline__ uint32_t valid15_starts(uint32_t m0, uint32_t m1)
{
    uint64_t M = ((uint64_t)m0 << 32) | m1;
    uint64_t A = M & (M << 1);
    A &= A << 2;
    A &= A << 4;
    A &= A << 7;
    return (uint32_t)(A >> 32);
}

// forward code of the 15-mer starting at base q (0..15) of word hi
__device__ __forceinline__ uint32_t k15_at(uint32_t hi, uint32_t lo, int q)
{
    if (q == 0) return hi >> 2;
    if (q == 1) return hi & K15_MASK;
    return __builtin_amdgcn_alignbit(hi, lo, 34 - 2 * q) & K15_MASK;
}

// reverse complement of a whole 16-base word: with R = (rc32(lo) : rc32(hi)) the reverse complement of the
// 15-mer starting at base q of (hi : lo) is (R >> 2q) & mask -- one v_alignbit per window instead of a bit reversal
__device__ __forceinline__ uint32_t rc32(uint32_t w)
{
    uint32_t r = __builtin_bitreverse32(w);
    r = ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
    return r ^ 0xAAAAAAAAu;
}

// pair index h of a 15-mer given both strands: the strand whose middle base has high code bit 0 is canonical
// (k is odd, complement = XOR 10b), and dropping that bit numbers the 2^29 pairs densely
__device__ __forceinline__ uint32_t cov_map_index_rc(uint32_t val, uint32_t rc)
{
    const uint32_t x = (val & 0x8000u) ? rc : val;
    return ((x >> 16) << 15) | (x & 0x7FFFu);
}

// coverage bin of a table count, kmer_utils.h:55-69
__device__ __forceinline__ uint32_t cov_bin_dev(uint32_t count, uint32_t bs, uint32_t bins)
{
    const uint32_t c = count < 2u ? 0u : count;
    if (c <= bs) return 0u;
    const uint32_t pos = c / bs - 1u;
    return (pos > 0u && pos < bins) ? pos : bins - 1u;
}

#endif
