// lrb_format.hip -- K8: the text form of the profile rows, written on the device.
//
// The reference prints every profile value with std::to_string(double) = "%f"
// (count-kmers.cpp:110-118, search-15mers.cpp:35-47).  Every value is a ratio in [0, 1], so
// "%f" is always eight characters ("0.dddddd" or "1.000000") and a row has a fixed width:
//   com_profs  9 * dim + 1 bytes  (a space after EVERY value, then the newline)
//   cov_profs  9 * bins bytes     (single spaces between the values, newline after the last)
// which makes the text a dense byte matrix a kernel can fill: one thread per value, rows staged
// in LDS 16 at a time and stored as 16-byte words.  The 136 M values of 1 M reads at k = 4 cost
// the host formatter 0.58 s on 32 threads (the largest share of that stage); here they are a few
// milliseconds and one 1.2 GB copy.
//
// "%f" exactly: glibc rounds the EXACT binary value of the double to six decimals, ties to even.
// v = m * 2^e with m < 2^53, so m * 10^6 < 2^73 and the rounding is done on 128-bit integers
// (same arithmetic as put_f_exact in lrb_host.cpp, which the tests pin against snprintf).
#include "lrb_device.h"

namespace {

#define FMT_ROWS 16 // rows per workgroup: 16 * width is a multiple of 16 bytes

// six-decimal integer of v in [0, 1] (anything else: the caller's flag is raised by q > 10^6)
__device__ __forceinline__ uint32_t fmt_q6(double v)
{
    const uint64_t bits = (uint64_t)__double_as_longlong(v);
    const int ex = (int)((bits >> 52) & 0x7FF);
    if ((bits >> 63) || ex >= 1023 + 20) return 0xFFFFFFFFu; // negative, >= 2^20, inf, nan
    uint64_t m = bits & ((1ull << 52) - 1);
    int e; // v = m * 2^e
    if (ex == 0) {
        e = -1074;
    } else {
        m |= 1ull << 52;
        e = ex - 1075;
    }
    const unsigned __int128 x = (unsigned __int128)m * 1000000u;
    uint64_t q;
    if (e >= 0) {
        q = (uint64_t)(x << e);
    } else {
        const int sh = -e;
        if (sh > 127) {
            q = 0;
        } else {
            const unsigned __int128 one = (unsigned __int128)1 << sh;
            const unsigned __int128 rem = x & (one - 1), half = one >> 1;
            q = (uint64_t)(x >> sh);
            if (rem > half || (rem == half && (q & 1))) ++q;
        }
    }
    return q > 0xFFFFFFFEull ? 0xFFFFFFFFu : (uint32_t)q;
}

// "d.dddddd" of q (<= 10^6) and the separator that follows, at p (LDS)
__device__ __forceinline__ void fmt_put(uint8_t *p, uint32_t q, uint8_t sep)
{
    const uint32_t ip = q / 1000000u;
    uint32_t fp = q - ip * 1000000u;
    p[0] = (uint8_t)('0' + ip);
    p[1] = '.';
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        const uint32_t t = fp / 10u;
        p[2 + i] = (uint8_t)('0' + (fp - t * 10u));
        fp = t;
    }
    p[8] = sep;
}

// MODE 0: com_profs (count-kmers.cpp:89-92,110-118)   value = c / max(1, L - k + 1), ' ' after each, then '\n'
// MODE 1: cov_profs (kmer_utils.h:74-84, search-15mers.cpp:35-47)
//         sum > 0: value = h / sum, below 1e-4 -> 0; sum == 0: the raw count; ' ' between, '\n' last
template <int MODE>
__global__ __launch_bounds__(256) void fmt_rows_kernel(const uint32_t *__restrict__ vals, const uint32_t *__restrict__ per_row,
                                                       uint64_t n, uint32_t dim, int k, uint8_t *__restrict__ text,
                                                       uint32_t *__restrict__ q_out, int *__restrict__ flag)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t tile[];
    const uint32_t width = MODE == 0 ? 9u * dim + 1u : 9u * dim;
    const uint64_t r0 = (uint64_t)blockIdx.x * FMT_ROWS;
    const uint32_t rows = n - r0 < FMT_ROWS ? (uint32_t)(n - r0) : FMT_ROWS;
    const uint32_t count = rows * dim;
    bool bad = false;
    for (uint32_t i = threadIdx.x; i < count; i += 256) {
        const uint32_t r = i / dim, j = i - r * dim;
        const uint32_t c = vals[(r0 + r) * dim + j], t = per_row[r0 + r];
        double v;
        if (MODE == 0) {
            const double total = t >= (uint32_t)k ? (double)(t - (uint32_t)k + 1u) : 0.0;
            v = (double)c / (total < 1.0 ? 1.0 : total);
        } else {
            v = (double)c;
            if (t > 0) {
                v /= (double)t;
                if (v < 1e-4) v = 0.0;
            }
        }
        uint32_t q = fmt_q6(v);
        if (q > 1000000u) {
            bad = true;
            q = 0;
        }
        if (q_out) q_out[(r0 + r) * dim + j] = q;
        fmt_put(tile + r * width + 9u * j, q, (MODE == 1 && j + 1 == dim) ? (uint8_t)'\n' : (uint8_t)' ');
    }
    if (MODE == 0 && threadIdx.x < rows) tile[threadIdx.x * width + 9u * dim] = '\n';
    if (bad) atomicOr(flag, 1);
    lrb_barrier();
    // the tile is one contiguous run of the output, starting on a 16-byte boundary
    const uint64_t base = r0 * width;
    const uint32_t bytes = rows * width, words = bytes / 16u;
    uint4 *dst = reinterpret_cast<uint4 *>(text + base);
    const uint4 *src = reinterpret_cast<const uint4 *>(tile);
    for (uint32_t w = threadIdx.x; w < words; w += 256) dst[w] = src[w];
    for (uint32_t b = words * 16u + threadIdx.x; b < bytes; b += 256) text[base + b] = tile[b];
}

template <int MODE>
int fmt_launch(lrb_ctx *c, const uint32_t *d_vals, const uint32_t *d_per_row, uint64_t n, uint32_t dim, int k, uint8_t *d_text,
               uint32_t *d_q)
{
    if (n == 0) return LRB_OK;
    HIP_TRY(hipSetDevice(c->device));
    void *d_flag;
    int rc = lrb_ws_get(c, 11, 64, &d_flag);
    if (rc != LRB_OK) return rc;
    HIP_TRY(hipMemsetAsync(d_flag, 0, 4, c->stream));
    const uint32_t width = MODE == 0 ? 9u * dim + 1u : 9u * dim;
    const size_t smem = (size_t)FMT_ROWS * width;
    static lrb_per_device_once raised[2];
    if (smem > 48 * 1024 && raised[MODE].need(c->device)) {
        HIP_TRY(hipFuncSetAttribute((const void *)fmt_rows_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    const uint64_t blocks = (n + FMT_ROWS - 1) / FMT_ROWS;
    ARG_TRY(blocks <= 0x7FFFFFFFull);
    hipLaunchKernelGGL(fmt_rows_kernel<MODE>, dim3((unsigned)blocks), dim3(256), smem, c->stream, d_vals, d_per_row, n, dim, k, d_text,
                       d_q, (int *)d_flag);
    HIP_TRY(hipGetLastError());
    int flag = 0;
    HIP_TRY(hipMemcpyAsync(&flag, d_flag, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (flag) {
        lrb_set_error("a profile value is outside [0, 1]: %s%s", MODE == 0 ? "a count above the window total" : "a bin above its row sum",
                      " (lrb_format_com / lrb_format_cov format such rows on the host)");
        return LRB_ERR_ARG;
    }
    return LRB_OK;
}

} // namespace

extern "C" uint64_t lrb_com_row_bytes(uint32_t dim) { return 9ull * dim + 1; }
extern "C" uint64_t lrb_cov_row_bytes(uint32_t bins) { return 9ull * bins; }

extern "C" int lrb_format_com_dev(lrb_ctx *c, const uint32_t *d_counts, const uint32_t *d_lens, uint64_t n, uint32_t dim, int k,
                                  uint8_t *d_text, uint32_t *d_q)
{
    ARG_TRY(c != nullptr && k >= 1 && dim >= 1 && dim <= 1024);
    HIP_TRY(hipSetDevice(c->device));
    ARG_TRY(n == 0 || (d_counts != nullptr && d_lens != nullptr && d_text != nullptr));
    ARG_TRY(((uintptr_t)d_text & 15) == 0);
    return fmt_launch<0>(c, d_counts, d_lens, n, dim, k, d_text, d_q);
}

extern "C" int lrb_format_cov_dev(lrb_ctx *c, const uint32_t *d_hist, const uint32_t *d_sums, uint64_t n, uint32_t bins,
                                  uint8_t *d_text, uint32_t *d_q)
{
    ARG_TRY(c != nullptr && bins >= 1 && bins <= 1024);
    HIP_TRY(hipSetDevice(c->device));
    ARG_TRY(n == 0 || (d_hist != nullptr && d_sums != nullptr && d_text != nullptr));
    ARG_TRY(((uintptr_t)d_text & 15) == 0);
    return fmt_launch<1>(c, d_hist, d_sums, n, bins, 0, d_text, d_q);
}
