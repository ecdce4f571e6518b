// search-15mers <table> <reads> <out> <bin_size> <bins> <threads>     (argv of search-15mers.cpp:124-136)
// -> <out>: one text row per read, the coverage histogram of its valid 15-mers normalised, values below
// 1e-4 zeroed, "%f" separated by single spaces (kmer_utils.h:24-87, search-15mers.cpp:35-47).  K3 + K8.
#include "lrb_bin_common.h"

int main(int argc, char **argv)
{
    if (argc < 7) {
        fprintf(stderr, "usage: %s <table> <reads> <out> <bin_size> <bins> <threads>\n", argv[0]);
        return 1;
    }
    const char *table_path = argv[1], *reads = argv[2], *out_path = argv[3];
    const long long bin_size = atoll(argv[4]);
    const int bins = atoi(argv[5]), threads = atoi(argv[6]);
    lrb_ctx *ctx = nullptr;
    if (lrb_ctx_create(lrb_device_from_env(), nullptr, 1, &ctx) != LRB_OK) return lrb_fail("device");
    void *table = nullptr;
    if (lrb_dev_alloc(ctx, 4 * LRB_K15_ENTRIES, &table) != LRB_OK) return lrb_fail("table");
    if (lrb_k15_read_file(ctx, (uint32_t *)table, table_path) != LRB_OK) return lrb_fail("table file");
    FILE *out = fopen(out_path, "wb");
    if (!out) {
        perror(out_path);
        return 1;
    }
    std::vector<uint8_t> text;
    const int rc = lrb_for_each_batch(
        reads, threads,
        [&](const uint8_t *seqs, const uint64_t *offs, uint64_t n) -> int {
            text.resize((size_t)(n * lrb_cov_row_bytes((uint32_t)bins)));
            if (lrb_cov_text_host(ctx, seqs, offs, n, (const uint32_t *)table, bin_size, bins, text.data(), nullptr) != LRB_OK)
                return lrb_fail("coverage");
            if (fwrite(text.data(), 1, text.size(), out) != text.size()) {
                perror(out_path);
                return 1;
            }
            return 0;
        },
        [&]() -> int {
            out = freopen(out_path, "wb", out);
            return out ? 0 : 1;
        });
    if (out && fclose(out) != 0) return 1;
    lrb_dev_free(ctx, table);
    lrb_ctx_destroy(ctx);
    return rc;
}
