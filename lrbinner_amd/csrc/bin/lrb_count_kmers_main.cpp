// count-kmers <reads> <out> <k> <threads>          (argv of count-kmers.cpp:195-198)
// -> <out>: one text row per read, count / max(1, L - k + 1) printed "%f" + ' ' after every value
// (count-kmers.cpp:89-92,110-118), truncated at the start, rows in input order.  K1 + K8 on the GPU.
#include <string.h>
#include <chrono>

#include "lrb_bin_common.h"

int main(int argc, char **argv)
{
    if (argc < 5) {
        fprintf(stderr, "usage: %s <reads> <out> <k> <threads>\n", argv[0]);
        return 1;
    }
    const char *reads = argv[1], *out_path = argv[2];
    const int k = atoi(argv[3]), threads = atoi(argv[4]);
    uint32_t dim = 0;
    if (lrb_kmer_dim(k, &dim) != LRB_OK) return lrb_fail("k");
    lrb_ctx *ctx = nullptr;
    if (lrb_ctx_create(lrb_device_from_env(), nullptr, 1, &ctx) != LRB_OK) return lrb_fail("device");
    FILE *out = fopen(out_path, "wb");
    if (!out) {
        perror(out_path);
        return 1;
    }
    std::vector<uint8_t> text;
    uint64_t total = 0;
    const bool timing = getenv("LRB_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    double t_gpu = 0, t_io = 0, t_last = now(), t_wait = 0;
    const int rc = lrb_for_each_batch(
        reads, threads,
        [&](const uint8_t *seqs, const uint64_t *offs, uint64_t n) -> int {
            const double t0 = now();
            t_wait += t0 - t_last; // since the previous batch was done: waiting for the parser pool
            text.resize((size_t)(n * lrb_com_row_bytes(dim)));
            if (lrb_kmer_text_host(ctx, seqs, offs, n, k, text.data(), nullptr) != LRB_OK) return lrb_fail("count");
            const double t1 = now();
            if (fwrite(text.data(), 1, text.size(), out) != text.size()) {
                perror(out_path);
                return 1;
            }
            total += n;
            t_last = now();
            t_gpu += t1 - t0;
            t_io += t_last - t1;
            return 0;
        },
        [&]() -> int {
            out = freopen(out_path, "wb", out);
            total = 0;
            return out ? 0 : 1;
        });
    if (out && fclose(out) != 0) return 1;
    const double t_loop = now();
    lrb_ctx_destroy(ctx);
    if (timing)
        fprintf(stderr, "count-kmers: batches %.3f s = parser wait %.3f + upload/pack/tally/format/download %.3f + write %.3f; teardown %.3f s\n",
                t_loop - t_begin, t_wait, t_gpu, t_io, now() - t_loop);
    if (rc == 0) printf("composition vectors of %llu reads\n", (unsigned long long)total);
    return rc;
}
