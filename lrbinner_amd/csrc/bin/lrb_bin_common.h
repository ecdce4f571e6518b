// Shared by the three drop-in executables: the process + file boundary of the reference
// (mbcclr_utils/runners_utils.py:78-105 starts mbcclr_utils/bin/{count-kmers,count-15mers,search-15mers}
// with os.system and reads the exit status; stdout is not parsed by anyone).  Same argv, same output
// files, exit status 0 / non-zero -- everything else goes through the C ABI of include/lrb_hip.h.
#ifndef LRB_BIN_COMMON_H
#define LRB_BIN_COMMON_H

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "lrb_hip.h"

#define LRB_CHUNK_BYTES (1ull << 26) /* byte range one parser thread turns into one batch */
#define LRB_MAX_PARSER_THREADS 32

static inline int lrb_fail(const char *what)
{
    fprintf(stderr, "%s: %s\n", what, lrb_last_error());
    return 1;
}

static inline int lrb_device_from_env()
{
    const char *e = getenv("LRB_DEVICE");
    return e ? atoi(e) : 0;
}

// every batch of the file, in order: the pool of parser threads for plain FASTA, the serial reader for
// gzip / FASTQ and for files the pool cannot cut (LRB_ERR_FORMAT: start over serially; the caller truncates
// its output in `restart`)
template <typename Batch, typename Restart>
static int lrb_for_each_batch(const char *path, int threads, Batch on_batch, Restart restart)
{
    // the reference gives `threads` to its OpenMP team; here it sizes the pool of parser threads, which feeds the
    // GPU with 32 and only loses beyond (1 M x 10 kb: 0.43 s of batches with 32 threads, 1.0 s with 256)
    if (threads < 1) threads = 1;
    if (threads > LRB_MAX_PARSER_THREADS) threads = LRB_MAX_PARSER_THREADS;
    lrb_preader *rd = nullptr;
    if (lrb_preader_open(path, threads, LRB_CHUNK_BYTES, &rd) != LRB_OK) return lrb_fail("open");
    for (;;) {
        const uint8_t *seqs = nullptr;
        const uint64_t *offs = nullptr;
        uint64_t n = 0;
        const int rc = lrb_preader_next(rd, &seqs, &offs, &n);
        if (rc == LRB_ERR_FORMAT) {
            lrb_preader_close(rd);
            if (restart() != 0) return 1;
            lrb_reader *sr = nullptr;
            if (lrb_reader_open(path, &sr) != LRB_OK) return lrb_fail("open");
            for (;;) {
                if (lrb_reader_next(sr, 1u << 17, 1ull << 29, &seqs, &offs, &n) != LRB_OK) {
                    lrb_reader_close(sr);
                    return lrb_fail("read");
                }
                if (n == 0) break;
                if (on_batch(seqs, offs, n) != 0) {
                    lrb_reader_close(sr);
                    return 1;
                }
            }
            lrb_reader_close(sr);
            return 0;
        }
        if (rc != LRB_OK) {
            lrb_preader_close(rd);
            return lrb_fail("read");
        }
        if (n == 0) break;
        if (on_batch(seqs, offs, n) != 0) {
            lrb_preader_close(rd);
            return 1;
        }
    }
    lrb_preader_close(rd);
    return 0;
}

#endif
