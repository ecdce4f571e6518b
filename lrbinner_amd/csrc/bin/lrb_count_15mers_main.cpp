// count-15mers <reads> <out> <threads>             (argv of count-15mers.cpp:101-103)
// -> <out>: u64 entry count (2^30) + 2^30 u32, T[x] = occurrences of x and of rc(x) over all valid
// 15-mers of the reads (kmer_utils.h:89-97,114-156).  K2 on the GPU as the pipeline's stage runs it: window lists tallied
// into the canonical half of the table (lrb_packed_k15_tally_half_many), expanded once at the end.
#include "lrb_bin_common.h"

int main(int argc, char **argv)
{
    if (argc < 4) {
        fprintf(stderr, "usage: %s <reads> <out> <threads>\n", argv[0]);
        return 1;
    }
    const char *reads = argv[1], *out_path = argv[2];
    const int threads = atoi(argv[3]);
    lrb_ctx *ctx = nullptr;
    if (lrb_ctx_create(lrb_device_from_env(), nullptr, 1, &ctx) != LRB_OK) return lrb_fail("device");
    void *table = nullptr, *half = nullptr;
    if (lrb_dev_alloc(ctx, 4 * LRB_K15_ENTRIES, &table) != LRB_OK) return lrb_fail("table");
    if (lrb_dev_alloc(ctx, 2 * LRB_K15_ENTRIES, &half) != LRB_OK) return lrb_fail("table");
    if (lrb_dev_memset(ctx, half, 0, 2 * LRB_K15_ENTRIES) != LRB_OK) return lrb_fail("table");
    // batches stay packed in HBM while a third of the free memory allows (the lists of a group and their scratch want as
    // much again), and go into the half in groups that share one partition of their windows
    std::vector<lrb_packed *> held;
    uint64_t held_bytes = 0, budget = 0, total_b = 0;
    lrb_dev_mem_info(ctx, &budget, &total_b);
    budget /= 3;
    auto flush = [&]() -> int {
        int r = LRB_OK;
        if (!held.empty()) r = lrb_packed_k15_tally_half_many(ctx, held.data(), held.size(), (uint32_t *)half);
        for (lrb_packed *p : held) lrb_packed_free(ctx, p);
        held.clear();
        held_bytes = 0;
        return r == LRB_OK ? 0 : lrb_fail("accumulate");
    };
    int rc = lrb_for_each_batch(
        reads, threads,
        [&](const uint8_t *seqs, const uint64_t *offs, uint64_t n) -> int {
            lrb_packed *p = nullptr;
            if (lrb_packed_create(ctx, seqs, offs, n, 0, &p) != LRB_OK) return lrb_fail("pack");
            uint64_t bytes = 0;
            lrb_packed_info(p, nullptr, &bytes);
            held.push_back(p);
            held_bytes += bytes;
            return held_bytes > budget ? flush() : 0;
        },
        [&]() -> int {
            for (lrb_packed *p : held) lrb_packed_free(ctx, p);
            held.clear();
            held_bytes = 0;
            return lrb_dev_memset(ctx, half, 0, 2 * LRB_K15_ENTRIES) == LRB_OK ? 0 : 1;
        });
    if (rc == 0) rc = flush();
    if (rc == 0 && lrb_k15_expand_half_dev(ctx, (const uint32_t *)half, (uint32_t *)table) != LRB_OK) rc = lrb_fail("expand");
    if (rc == 0 && lrb_k15_write_file(ctx, (const uint32_t *)table, out_path) != LRB_OK) rc = lrb_fail("write");
    lrb_dev_free(ctx, half);
    lrb_dev_free(ctx, table);
    lrb_ctx_destroy(ctx);
    return rc;
}
