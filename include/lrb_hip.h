/*
 * lrb_hip.h -- C ABI of the MI355X (gfx950) profile / clustering kernels that
 * replace LRBinner's native hot path.
 *
 * The reference crosses this boundary as three executables started with
 * os.system (mbcclr_utils/runners_utils.py:78-105) that exchange files under
 * {output}/profiles/, plus two torch-CPU helpers in cluster_utils.py.  A
 * maintainer who wants the GPU path binds THIS library instead (ctypes stub in
 * INTEGRATION.md); every entry point below names the reference code it
 * replaces.  Plain C types only: pointers, sizes, an opaque context.
 *
 * Conventions
 *   - every function returns 0 (LRB_OK) or an LRB_ERR_* code and never throws
 *     or aborts across the boundary; lrb_last_error() gives the message of the
 *     last failure on the calling thread.
 *   - "d_" parameters are device pointers (hipMalloc / torch storage); all
 *     others are host pointers.  *_dev functions enqueue on the context's HIP
 *     stream and return without synchronising; *_host functions are
 *     synchronous (H2D, kernels, D2H).
 *   - reads are passed as concatenated bytes + uint64 offsets[n+1].
 *   - base code = (ascii >> 1) & 3  (A0 C1 T2 G3), count-kmers.cpp:77.
 *
 * Packed read layout in HBM (produced by lrb_pack_reads_dev, consumed by the
 * profile kernels; see DESIGN.md "Data layout"):
 *   codes : 2 bits per base, 16 bases per uint32, first base in bits 31..30.
 *           EVERY byte is coded (N, lowercase included) because the
 *           composition path never tests validity (count-kmers.cpp:73-87).
 *   mask  : 1 bit per base, 32 bases per uint32, first base in bit 31; the
 *           bit is 1 iff the byte is one of 'A','C','G','T' -- the validity
 *           rule of the 15-mer paths (kmer_utils.h:38-43,122-127).
 *   Read r starts at word code_off[r] of codes and mask_off[r] of mask; both
 *   regions are padded with zero words (lrb_pack_layout gives the sizes).
 */
#ifndef LRB_HIP_H
#define LRB_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LRB_OK 0
#define LRB_ERR_ARG 1     /* invalid argument (null pointer, k outside 3..5, ...) */
#define LRB_ERR_HIP 2     /* a HIP runtime call failed */
#define LRB_ERR_NOMEM 3   /* host or device allocation failed */
#define LRB_ERR_NODEVICE 4
#define LRB_ERR_IO 5      /* file could not be opened / read / written */
#define LRB_ERR_FORMAT 6  /* malformed input file */

#define LRB_K15_ENTRIES 1073741824ull /* 4^15 slots of the 15-mer table */
#define LRB_HIST_BINS 60              /* ceil(_XMAX/_DELTA_X), cluster_utils.py:52-53,138 */

typedef struct lrb_ctx lrb_ctx;

/* ---- library / context ------------------------------------------------ */
const char *lrb_last_error(void);
int lrb_version(void);
int lrb_device_count(int *count);

/* own_stream != 0: the context creates (and later destroys) its own non-blocking
 * HIP stream and `stream` is ignored.  own_stream == 0: every launch goes to the
 * caller's hipStream_t `stream` (e.g. torch's current stream; NULL is the
 * legacy default stream). */
int lrb_ctx_create(int device, void *stream, int own_stream, lrb_ctx **out);
int lrb_ctx_destroy(lrb_ctx *ctx);
int lrb_ctx_sync(lrb_ctx *ctx);
/* Give back the context's grow-on-demand workspaces of keep_below bytes and more (the partition buffers
 * of a K2 group are 6 bytes per window, up to 26 GB): after the stream has drained.  They come back on
 * the next call that needs them. */
int lrb_ctx_trim(lrb_ctx *ctx, uint64_t keep_below);
/* Long-lived hosts that keep slice lists of their own across the collective (lrb_packed_lists_create with
 * in_workspace = 0), call after call: up to max_bytes of the memory lrb_winlists_free gives back is retained by the
 * context and handed to the next lists of (nearly) the same size instead of going through hipFree / hipMalloc --
 * a 17.6 GB hipMalloc was measured at 0.13-0.48 s on MI355X once ~50 GB are allocated, thirty times the partition
 * pass the kept lists save (profiles/r05_side_alloc.txt).  0 (the default): nothing is retained.  lrb_ctx_trim
 * frees retained blocks of keep_below bytes and more. */
int lrb_ctx_list_pool(lrb_ctx *ctx, uint64_t max_bytes);
/* Diagnostics: where the context's workspace slot `slot` lives right now (device address, 0 when never used) and its
 * size; the contents are whatever the last call that used the slot left (slot 9: the level-1 window lists of the last
 * partition).  Valid until the next call that uses the slot.  scripts/k2_stress3.py. */
int lrb_ctx_ws_info(const lrb_ctx *ctx, int slot, void **d_ptr, uint64_t *bytes);
/* How often this context has REPEATED a partition of window lists because the part kernel appended to some (unit, slice)
 * another number of windows than the count kernel had counted for it -- the same arithmetic on the same words, so a
 * disagreement is an execution fault, not the data's.  Round 6 saw it once in about a hundred partitions when eight
 * processes were time-sliced on ONE GPU (a barrier without a wait for the waves' LDS atomics: fixed at every barrier of
 * the library since), never with a GPU to itself; the check stays as a guard.  Undetected such a fault loses a
 * window of kmer_utils.h:114-156's table; detected, the partition is made again (five attempts, then LRB_ERR_HIP). */
int lrb_ctx_partition_retries(const lrb_ctx *ctx, uint64_t *count);
int lrb_ctx_stream(lrb_ctx *ctx, void **stream);

/* plain device memory helpers for callers without torch */
int lrb_dev_alloc(lrb_ctx *ctx, uint64_t bytes, void **d_ptr);
int lrb_dev_free(lrb_ctx *ctx, void *d_ptr);
/* Free / total device memory as hipMemGetInfo reports it: what a caller sizes its resident
 * batches by (the reference streams the file through 10,000-read batches, count-kmers.cpp:141). */
int lrb_dev_mem_info(lrb_ctx *ctx, uint64_t *free_bytes, uint64_t *total_bytes);
/* Page-locked host memory for result buffers that are filled over and over (the text rows of
 * lrb_packed_*_text): copies into it run at link speed and skip the first-touch page faults of a
 * fresh allocation. */
int lrb_host_alloc(lrb_ctx *ctx, uint64_t bytes, void **h_ptr);
int lrb_host_free(lrb_ctx *ctx, void *h_ptr);
int lrb_dev_memset(lrb_ctx *ctx, void *d_ptr, int value, uint64_t bytes);
int lrb_copy_h2d(lrb_ctx *ctx, void *d_dst, const void *src, uint64_t bytes);
int lrb_copy_d2h(lrb_ctx *ctx, void *dst, const void *d_src, uint64_t bytes);

/* ---- canonical k-mer index ------------------------------------------- */
/* Replaces compute_kmer_inds (count-kmers.cpp:38-64).  k in {3,4,5};
 * lut has 4^k entries; *dim = 32 / 136 / 512. */
int lrb_kmer_dim(int k, uint32_t *dim);
int lrb_kmer_lut(int k, uint32_t *lut, uint32_t *dim);

/* ---- packed layout ---------------------------------------------------- */
/* From byte offsets[n+1] compute lens[n], code_off[n+1], mask_off[n+1] (in
 * uint32 words).  code_off[n] / mask_off[n] are the array sizes to allocate.
 * Fails with LRB_ERR_ARG for a read of 2^32 bases or more. */
int lrb_pack_layout(const uint64_t *offs, uint64_t n, uint32_t *lens,
                    uint64_t *code_off, uint64_t *mask_off);

/* ASCII -> packed codes (+ validity mask, + bit planes; d_mask / d_planes may be NULL).
 * Replaces the per-byte coding inside count_kmers / line_to_vec / line_to_kmer_counts.
 * seq_bytes = offs[n] (size of d_seqs).  d_planes: see lrb_kmer_counts3_dev. */
int lrb_pack_reads_dev(lrb_ctx *ctx, const uint8_t *d_seqs, uint64_t seq_bytes,
                       const uint64_t *d_offs, uint64_t n,
                       const uint64_t *d_code_off, const uint64_t *d_mask_off,
                       uint32_t *d_codes, uint32_t *d_mask, uint32_t *d_planes);

/* ---- K1: composition -------------------------------------------------- */
/* Integer view of count_kmers (count-kmers.cpp:66-87): d_counts[r*dim + c] =
 * number of windows of read r whose canonical index is c.  The host divides
 * by max(1, len-k+1) (count-kmers.cpp:89-92). */
int lrb_kmer_counts_dev(lrb_ctx *ctx, const uint32_t *d_codes, const uint64_t *d_code_off,
                        const uint32_t *d_lens, uint64_t n, int k, uint32_t *d_counts);
int lrb_kmer_counts_host(lrb_ctx *ctx, const uint8_t *seqs, const uint64_t *offs, uint64_t n,
                         int k, uint32_t *counts);

/* k = 3 on the bit-plane form of the reads.  d_planes holds, for block b (32 bases)
 * of read r, the pair {H, L} at words 2*(mask_off[r]+b), 2*(mask_off[r]+b)+1: H = the
 * high bit of each base code, L = the low bit, first base in bit 31; same padding as
 * the mask.  lrb_planes_from_codes_dev derives it from `codes`.
 * Per-read planes are the INPUT of the group-transposed layout (lrb_planes_t_from_planes_dev -> lrb_kmer_counts3t_dev, the
 * k = 3 kernel).  lrb_kmer_counts3_dev itself tallies from d_codes whatever the mode (0, 1, 2 accepted: the wave-per-read
 * bit-plane kernel that modes 0 / 2 used to run went in round 5); same result as lrb_kmer_counts_dev(..., k = 3, ...). */
int lrb_planes_from_codes_dev(lrb_ctx *ctx, const uint32_t *d_codes, const uint64_t *d_code_off,
                              const uint64_t *d_mask_off, uint64_t n, uint32_t *d_planes);
int lrb_kmer_counts3_dev(lrb_ctx *ctx, const uint32_t *d_codes, const uint32_t *d_planes,
                         const uint64_t *d_code_off, const uint64_t *d_mask_off,
                         const uint32_t *d_lens, uint64_t n, int mode, uint32_t *d_counts);

/* k = 3 on the GROUP-TRANSPOSED bit planes (lane-per-read kernel, the fast path).
 * Reads are taken in groups of 64: slot s = 64*g + lane holds read order[s] (order NULL:
 * read s).  Row j of group g holds block j of its 64 reads:
 *   d_planes_t[((group_off[g] + j) * 64 + lane) * 2 + {0: H, 1: L}],
 * group_off[g+1] - group_off[g] = 1 + the group's largest block count (last row = zero
 * halo).  lrb_planes_t_layout fills group_off[(n+63)/64 + 1] on the host and, when order
 * is given, order[n] = the reads sorted by length (a group then costs its longest read
 * and nothing idles).  2 * 64 * group_off[last] uint32 is the size of d_planes_t.
 * lrb_pack_planes_t_dev writes the layout straight from ASCII;
 * lrb_planes_t_from_planes_dev converts the per-read planes. */
int lrb_planes_t_layout(const uint32_t *lens, uint64_t n, uint32_t *order, uint64_t *group_off);
int lrb_pack_planes_t_dev(lrb_ctx *ctx, const uint8_t *d_seqs, const uint64_t *d_offs,
                          const uint64_t *d_group_off, const uint32_t *d_order, uint64_t n,
                          uint32_t *d_planes_t);
int lrb_planes_t_from_planes_dev(lrb_ctx *ctx, const uint32_t *d_planes, const uint64_t *d_mask_off,
                                 const uint64_t *d_group_off, const uint32_t *d_order, uint64_t n,
                                 uint32_t *d_planes_t);
int lrb_kmer_counts3t_dev(lrb_ctx *ctx, const uint32_t *d_planes_t, const uint64_t *d_group_off,
                          const uint32_t *d_order, const uint32_t *d_lens, uint64_t n,
                          uint32_t *d_counts);

/* k = 4 on GROUP-TRANSPOSED 2-bit codes (lane-per-read kernel with private LDS histograms, the
 * fast path for the composition width of BASELINE configs 3-5; count_kmers, count-kmers.cpp:66-87).
 * Groups of 64 reads as above; row j of group g holds code words 4j..4j+3 (64 bases) of its reads:
 *   d_codes_t[((group_off[g] + j) * 64 + lane) * 4 + word],
 * group_off[g+1] - group_off[g] = 1 + the group's largest row count (last row = zero halo), rows
 * counted in units of 64 bases (lrb_codes_t_layout; same arguments as lrb_planes_t_layout).
 * 4 * 64 * group_off[last] uint32 is the size of d_codes_t.  d_counts[n][136], uint32, rows in
 * read order whatever `order` is. */
int lrb_codes_t_layout(const uint32_t *lens, uint64_t n, uint32_t *order, uint64_t *group_off);
int lrb_codes_t_from_codes_dev(lrb_ctx *ctx, const uint32_t *d_codes, const uint64_t *d_code_off,
                               const uint64_t *d_group_off, const uint32_t *d_order, uint64_t n,
                               uint32_t *d_codes_t);
int lrb_kmer_counts4t_dev(lrb_ctx *ctx, const uint32_t *d_codes_t, const uint64_t *d_group_off,
                          const uint32_t *d_order, const uint32_t *d_lens, uint64_t n,
                          uint32_t *d_counts);
/* The same kernel for k = 4 or 5 on the same layout (k = 5: half a group per workgroup, 1024 bins x 32
 * columns = 64 KB of LDS, two 8-wave workgroups per CU; d_counts[n][512]) -- and k = 3 (4-mers at even
 * positions, 256 bins, folded into the 32 classes; d_counts[n][32]): the same tallies as lrb_kmer_counts3t_dev
 * at the same speed, for hosts that keep one layout for every k. */
int lrb_kmer_counts_t_dev(lrb_ctx *ctx, int k, const uint32_t *d_codes_t, const uint64_t *d_group_off,
                          const uint32_t *d_order, const uint32_t *d_lens, uint64_t n,
                          uint32_t *d_counts);

/* ---- K2: global 15-mer table ------------------------------------------ */
/* line_to_kmer_counts (kmer_utils.h:114-156) split in two linear steps:
 *   accumulate: F[val] += 1 for every valid 15-mer (forward code only)
 *   mirror    : T[x] = F[x] + F[rc(x)]  (uint32 wrap), in place
 * which equals the reference's T[val]++, T[rc(val)]++.  Accumulate may be
 * called for many batches (and on many GPUs, summed with an all-reduce)
 * before ONE mirror.  d_table has LRB_K15_ENTRIES uint32. */
int lrb_k15_accumulate_dev(lrb_ctx *ctx, const uint32_t *d_codes, const uint32_t *d_mask,
                           const uint64_t *d_code_off, const uint64_t *d_mask_off,
                           const uint32_t *d_lens, uint64_t n, uint32_t *d_table);
/* The same call under its older name (max_windows is ignored).  Rounds 1-3 ran a radix-partitioned
 * forward accumulate here; since round 5 the fast K2 is the CANONICAL-HALF route -- window lists
 * (lrb_k15_lists_part_dev / _tally_dev, lrb_packed_k15_tally_half_many) -- and forward tallies are for
 * callers that need F itself (tests, the full-table all-reduce A/B). */
int lrb_k15_accumulate_part_dev(lrb_ctx *ctx, const uint32_t *d_codes, const uint32_t *d_mask,
                                const uint64_t *d_code_off, const uint64_t *d_mask_off,
                                const uint32_t *d_lens, uint64_t n, uint64_t max_windows,
                                uint32_t *d_table);
int lrb_k15_mirror_dev(lrb_ctx *ctx, uint32_t *d_table);
/* The mirror step for a table that is the SUM over several GPUs: the result is determined
 * by its canonical half -- of x and rc(x) the one whose middle base has high code bit 0 (bit 15 of
 * x), numbered densely by dropping that bit: h = ((x >> 16) << 15) | (x & 0x7FFF), 2^29 uint32 =
 * 2 GiB.  fold: d_half[h] = F[x] + F[rc(x)] from a rank's forward tallies; the caller all-reduces
 * d_half (half the bytes of the table); expand: d_table[x] = d_table[rc(x)] = d_half[h].
 * expand(allreduce(fold(F_r))) == mirror(allreduce(F_r)) bit for bit (uint32 wrap). */
#define LRB_K15_HALF_ENTRIES 536870912ull
int lrb_k15_fold_half_dev(lrb_ctx *ctx, const uint32_t *d_table, uint32_t *d_half);
int lrb_k15_expand_half_dev(lrb_ctx *ctx, const uint32_t *d_half, uint32_t *d_table);
int lrb_k15_accumulate_host(lrb_ctx *ctx, const uint8_t *seqs, const uint64_t *offs,
                            uint64_t n, uint32_t *d_table);
/* writeKmerFile / readKmerFile (kmer_utils.h:89-112): u64 entry count + raw u32. */
int lrb_k15_write_file(lrb_ctx *ctx, const uint32_t *d_table, const char *path);
/* The same on a thread and a stream of the library's own, so that the caller can go on while
 * 4 GiB travel to the file (written as path.partial, renamed when complete).  d_table must stay
 * allocated and unchanged until lrb_job_wait, which returns the writer's status and frees the job. */
typedef struct lrb_job lrb_job;
int lrb_k15_write_file_async(lrb_ctx *ctx, const uint32_t *d_table, const char *path, lrb_job **job);
/* ONE part of the table file, for writers that share the work (every rank of the sharded driver holds the whole table
 * after the all-reduce): entries [part E / n_parts, (part + 1) E / n_parts) go to their place in the EXISTING file
 * `path` (made at its full size, 8 + 4 E bytes, by one of the writers; no rename here -- the callers agree on when the
 * file is complete); part 0 also writes the entry count in front.  lrb_job_wait as above. */
int lrb_k15_write_file_part_async(lrb_ctx *ctx, const uint32_t *d_table, const char *path, uint32_t part, uint32_t n_parts,
                                  lrb_job **job);
int lrb_job_wait(lrb_job *job);
int lrb_k15_read_file(lrb_ctx *ctx, uint32_t *d_table, const char *path);

/* ---- the collective: sum of the table over the GPUs of a node ------------- */
/* reads shard across ranks, the 15-mer table is a sum over reads, so the path has exactly
 * one exchange step.  lrb_k15_allreduce sums `count` uint32 (wrap-around) in place over all ranks of
 * an RCCL communicator, enqueued on the context's stream: the canonical half after
 * lrb_k15_fold_half_dev (count = LRB_K15_HALF_ENTRIES) or the whole forward table
 * (LRB_K15_ENTRIES).  `rccl_comm` is an ncclComm_t -- the host's own, or one made here:
 * rank 0 calls lrb_rccl_unique_id, hands the 128 bytes to the other ranks by whatever means it has
 * (MPI, a socket, a file), and every rank calls lrb_rccl_comm_create.  One process per GPU.
 * RCCL is bound at run time; LRB_ERR_NODEVICE when it is not installed. */
#define LRB_RCCL_ID_BYTES 128
int lrb_rccl_unique_id(uint8_t *id);
int lrb_rccl_comm_create(lrb_ctx *ctx, int n_ranks, int rank, const uint8_t *id, void **rccl_comm);
int lrb_rccl_comm_destroy(void *rccl_comm);
int lrb_k15_allreduce(lrb_ctx *ctx, void *rccl_comm, uint32_t *d_buf, uint64_t count);

/* ---- K3: coverage histogram ------------------------------------------- */
/* Integer view of line_to_vec (kmer_utils.h:24-72): d_hist[r*bins + b] and
 * d_sums[r] = number of valid 15-mers of read r.  The host normalises and
 * zeroes values < 1e-4 (kmer_utils.h:74-84).  bin_size >= 1, 1 <= bins <= 1024. */
int lrb_cov_hist_dev(lrb_ctx *ctx, const uint32_t *d_codes, const uint32_t *d_mask,
                     const uint64_t *d_code_off, const uint64_t *d_mask_off,
                     const uint32_t *d_lens, uint64_t n, const uint32_t *d_table,
                     int64_t bin_size, int bins, uint32_t *d_hist, uint32_t *d_sums);
/* The same histograms from a COMPACT MAP of the finished table: line_to_vec needs only the bin of a
 * count, and T[x] == T[rc(x)] after the mirror step, so lrb_cov_map_build_dev writes one byte per
 * pair (x, rc(x)) -- d_map[h] = bin of T[x], h as in lrb_k15_fold_half_dev, 2^29 bytes, bins <= 256 --
 * and lrb_cov_hist_map_dev gathers bytes from 512 MB (half of it Infinity-Cache resident) instead of
 * words from 4 GiB.  The map belongs to (table contents, bin_size, bins): rebuild it when any changes. */
#define LRB_COV_MAP_BYTES 536870912ull
int lrb_cov_map_build_dev(lrb_ctx *ctx, const uint32_t *d_table, int64_t bin_size, int bins, uint8_t *d_map);
int lrb_cov_hist_map_dev(lrb_ctx *ctx, const uint32_t *d_codes, const uint32_t *d_mask,
                         const uint64_t *d_code_off, const uint64_t *d_mask_off,
                         const uint32_t *d_lens, uint64_t n, const uint8_t *d_map, int bins,
                         uint32_t *d_hist, uint32_t *d_sums);
/* The same histograms again, as a SWEEP: the windows of the reads are brought to the map instead of the map to the
 * windows (window lists, next section: lrb_k15_lists_part_dev, then lrb_cov_lists_sweep_dev) -- no 128-byte line
 * fill per gather (several times lrb_cov_hist_map_dev from ~0.1 M reads of 10 kb on; below ~15 k reads use the
 * gather form).  Workspace: 128 bytes per mask word for the lists (context slot 8) and as much again for the part
 * kernel's scratch (slot 9), each at most 24 GB or a third / a quarter of the free memory: a larger batch is swept
 * in ranges of reads.  The call reads d_mask_off[0] and d_mask_off[n] back (stream synchronisations) to size it. */
int lrb_cov_hist_sweep_dev(lrb_ctx *ctx, const uint32_t *d_codes, const uint32_t *d_mask,
                           const uint64_t *d_code_off, const uint64_t *d_mask_off,
                           const uint32_t *d_lens, uint64_t n, const uint8_t *d_map, int bins,
                           uint32_t *d_hist, uint32_t *d_sums);
int lrb_cov_hist_host(lrb_ctx *ctx, const uint8_t *seqs, const uint64_t *offs, uint64_t n,
                      const uint32_t *d_table, int64_t bin_size, int bins,
                      uint32_t *hist, uint32_t *sums);

/* ---- K2 and K3 on ONE partition of the windows (WINDOW LISTS; round 4 form) ------------------
 * line_to_kmer_counts (kmer_utils.h:114-156) and line_to_vec (kmer_utils.h:24-87) walk the same reads and
 * extract the same 15-mers; count-15mers and search-15mers each parse the file for it.  Here the windows of a
 * set of resident reads are partitioned ONCE -- per GROUP of reads_per_group reads, by the top 14 bits of the
 * pair index (x and rc(x) share one, h as in lrb_k15_fold_half_dev): 16,384 BUCKETS of 2^15 pairs = 32 KB of the
 * compact map -- into lists of {read in the group : 11 | h & (2^21 - 1) : 21}, and both stages start from them:
 *   lrb_k15_lists_part_dev    the lists of every group (d_lists), where each bucket starts in its group's region
 *                             (d_bounds[g][0..16384]) and where the regions start (d_gbase[g], list slots)
 *   lrb_k15_lists_tally_dev   a workgroup per bucket, the bucket's 2^15 counters in LDS, coalesced add into the
 *                             CANONICAL HALF d_half[2^29]: d_half[h] = number of valid 15-mers of the reads whose pair
 *                             index is h.  That IS the reference's table: T[x] = T[rc(x)] = sum over ranks of
 *                             d_half[h(x)] (lrb_k15_expand_half_dev writes T out; the ranks all-reduce d_half as it
 *                             stands).
 *   lrb_cov_lists_sweep_dev   K3: a workgroup per group, its histograms in LDS, the buckets walked in order with the
 *                             bucket's 32 KB of the map staged in LDS (d_map: lrb_cov_map_build(_half)_dev)
 *   lrb_cov_map_build_half_dev  the compact map from d_half instead of the mirrored table (same bytes)
 *   lrb_k15_accumulate_half_dev one atomic per window into d_half (small batches)
 * Buffers are the caller's: d_lists = 32 uint32 per mask word of the reads (d_mask_off[n] - d_mask_off[0] words),
 * d_bounds = lrb_k15_lists_bounds_words(n_groups) uint32, d_gbase = n_groups + 1 uint64.  reads_per_group / the number
 * of groups for n reads and a histogram of `bins` bins: lrb_k15_lists_geometry (bins * reads_per_group <= 65,024: the
 * group's u16 counters beside the staged bucket in 160 KB of LDS; at most 2,048).  The part call uses context scratch
 * (slots 9, 10; the groups go through it in chunks when it is smaller than the lists) and synchronises the stream
 * once.  Reads of more than 65,535 windows are not in the lists; both consumers handle them by gathers / atomics. */
int lrb_k15_lists_geometry(lrb_ctx *ctx, uint64_t n, int bins, uint32_t *reads_per_group, uint64_t *n_groups);
/* ... with the reads' total length known (0 = unknown): groups small enough that a group's windows of one 2 MB slice
 * (1/256 of its windows on average) fit the 65,536 entries the order kernel holds in registers */
int lrb_k15_lists_geometry_for(lrb_ctx *ctx, uint64_t n, uint64_t total_bases, int bins, uint32_t *reads_per_group,
                               uint64_t *n_groups);
uint64_t lrb_k15_lists_bounds_words(uint64_t n_groups);
int lrb_k15_lists_part_dev(lrb_ctx *ctx, const uint32_t *d_codes, const uint32_t *d_mask,
                           const uint64_t *d_code_off, const uint64_t *d_mask_off, const uint32_t *d_lens,
                           uint64_t n, uint32_t reads_per_group, uint32_t *d_lists, uint32_t *d_bounds,
                           uint64_t *d_gbase);
int lrb_k15_lists_tally_dev(lrb_ctx *ctx, const uint32_t *d_codes, const uint32_t *d_mask,
                            const uint64_t *d_code_off, const uint64_t *d_mask_off, const uint32_t *d_lens,
                            uint64_t n, uint32_t reads_per_group, const uint32_t *d_lists,
                            const uint32_t *d_bounds, const uint64_t *d_gbase, uint32_t *d_half);
int lrb_k15_accumulate_half_dev(lrb_ctx *ctx, const uint32_t *d_codes, const uint32_t *d_mask,
                                const uint64_t *d_code_off, const uint64_t *d_mask_off,
                                const uint32_t *d_lens, uint64_t n, uint32_t *d_half);
int lrb_cov_map_build_half_dev(lrb_ctx *ctx, const uint32_t *d_half, int64_t bin_size, int bins, uint8_t *d_map);
int lrb_cov_lists_sweep_dev(lrb_ctx *ctx, const uint32_t *d_codes, const uint32_t *d_mask,
                            const uint64_t *d_code_off, const uint64_t *d_mask_off, const uint32_t *d_lens,
                            uint64_t n, uint32_t reads_per_group, const uint32_t *d_lists,
                            const uint32_t *d_bounds, const uint64_t *d_gbase, const uint8_t *d_map, int bins,
                            uint32_t *d_hist, uint32_t *d_sums);

/* ---- resident batches --------------------------------------------------- */
/* The reference parses the reads file once per binary (three times per run).  A
 * resident batch is uploaded and packed ONCE and stays in HBM; the three stages then
 * run on it without touching the file or PCIe again.  Results are host buffers as in
 * the *_host functions.  with_planes: bit 0 also keep the group-transposed bit planes (k = 3
 * kernel), bit 1 the group-transposed codes (k = 4, 5 kernel); without them the composition
 * of that k runs on the per-read codes (the wave-per-read LDS kernel). */
typedef struct lrb_packed lrb_packed;
int lrb_packed_create(lrb_ctx *ctx, const uint8_t *seqs, const uint64_t *offs, uint64_t n,
                      int with_planes, lrb_packed **out);
/* The same when the bases are ALREADY in HBM (d_seqs: device address of the byte offs[] index into; offs stays a
 * host array): only the offsets and lengths cross PCIe.  For hosts that read storage straight into device memory,
 * and for bench.py, whose synthetic reads are made on the device. */
int lrb_packed_create_dev(lrb_ctx *ctx, const uint8_t *d_seqs, const uint64_t *offs, uint64_t n,
                          int with_planes, lrb_packed **out);
/* ... and when the reads arrive ALREADY PACKED by the host (lrb_pack_reads_host, or the parser pool's packed view:
 * lrb_preader_open_ex with LRB_PREADER_PACKED): codes, masks, offsets and lengths are uploaded as they are (0.375 bytes a
 * base over PCIe), the transposed layouts are made from the codes on the device.  Same batch, bit for bit. */
int lrb_packed_create_packed(lrb_ctx *ctx, const uint32_t *codes, const uint32_t *mask, const uint64_t *code_off,
                             const uint64_t *mask_off, const uint32_t *lens, const uint64_t *offs, uint64_t n,
                             int with_planes, lrb_packed **out);
int lrb_packed_free(lrb_ctx *ctx, lrb_packed *p);
int lrb_packed_info(const lrb_packed *p, uint64_t *n, uint64_t *device_bytes);
int lrb_packed_kmer_counts(lrb_ctx *ctx, const lrb_packed *p, int k, uint32_t *counts);
/* ... its kernel half: the tallies stay in HBM (d_counts: n x dim uint32, device memory). */
int lrb_packed_kmer_counts_dev(lrb_ctx *ctx, const lrb_packed *p, int k, uint32_t *d_counts);
/* ... of `count` resident batches, rows in batch order (d_counts: sum of n x dim uint32).  k = 4 on batches that hold
 * the group-transposed codes: ONE launch over all their groups instead of one per batch (count-kmers.cpp:66-95 per read;
 * a batch of the parser pool is too few reads to fill the chip); anything else is the per-batch call in turn. */
int lrb_packed_kmer_counts_many_dev(lrb_ctx *ctx, const lrb_packed *const *packs, uint64_t count, int k,
                                    uint32_t *d_counts);
int lrb_packed_k15_accumulate(lrb_ctx *ctx, const lrb_packed *p, uint32_t *d_table);
/* The same for count resident batches at once: the partitioned accumulate passes over the whole
 * table once per call, so batches are grouped (up to 2^31 windows per group,
 * LRB_K2_GROUP_WINDOWS) and a group shares that pass.  Same table as count single calls. */
int lrb_packed_k15_accumulate_many(lrb_ctx *ctx, const lrb_packed *const *ps, uint64_t count,
                                   uint32_t *d_table);
int lrb_packed_cov_hist(lrb_ctx *ctx, const lrb_packed *p, const uint32_t *d_table,
                        int64_t bin_size, int bins, uint32_t *hist, uint32_t *sums);
/* The same two stages ending in the TEXT rows of the profile files, formatted on the device
 * (K8 below): text = n * lrb_com_row_bytes(dim) / n * lrb_cov_row_bytes(bins) host bytes, ready to
 * be appended to profiles/com_profs / profiles/cov_profs; q6 (optional, n * dim / n * bins) = the
 * six-decimal integer of every value -- the text parses to q6 / 1e6, which is what
 * pipelines.py:315-321 puts into the .npy files. */
int lrb_packed_kmer_text(lrb_ctx *ctx, const lrb_packed *p, int k, uint8_t *text, uint32_t *q6);
/* ... and for a host batch that is not kept resident (upload, pack, tally, format in the context's
 * workspaces): what the drop-in executables call per batch. */
int lrb_kmer_text_host(lrb_ctx *ctx, const uint8_t *seqs, const uint64_t *offs, uint64_t n, int k,
                       uint8_t *text, uint32_t *q6);
int lrb_cov_text_host(lrb_ctx *ctx, const uint8_t *seqs, const uint64_t *offs, uint64_t n,
                      const uint32_t *d_table, int64_t bin_size, int bins, uint8_t *text, uint32_t *q6);
int lrb_packed_cov_text(lrb_ctx *ctx, const lrb_packed *p, const uint32_t *d_table,
                        int64_t bin_size, int bins, uint8_t *text, uint32_t *q6);
/* K3 of MANY resident batches as ONE sweep against a compact map (lrb_cov_map_build_dev): the batches' packed
 * reads are taken as one batch (masks and lengths laid end to end in workspace, the codes reached where the batches
 * hold them) and lrb_cov_hist_sweep_dev runs on the lot -- one reader batch is too few
 * reads for it.  The histograms stay in the context, rows in batch order, until the next K1 / K3 call of the
 * context; lrb_cov_rows_text formats rows [first_row, first_row + n_rows) of them as lrb_packed_cov_text does
 * (text: n_rows * lrb_cov_row_bytes(bins) bytes; q6 optional).  bins <= 256. */
int lrb_packed_cov_hist_many(lrb_ctx *ctx, const lrb_packed *const *packs, uint64_t count, const uint8_t *d_map,
                             int bins);
int lrb_cov_rows_text(lrb_ctx *ctx, uint64_t first_row, uint64_t n_rows, int bins, uint8_t *text, uint32_t *q6);

/* The windows of MANY resident batches partitioned ONCE for both 15-mer stages (lrb_k15_lists_part_dev on the batches
 * laid end to end; at most 2^32 - 1 bases in all): lrb_winlists_tally adds their tallies to the canonical half of the
 * table (K2), lrb_winlists_cov_hist sweeps the same lists against a compact map (K3; the histograms stay in the context
 * for lrb_cov_rows_text, rows in batch order; bins * reads_per_group <= 65,024, which lists made for `bins` meet
 * for any smaller histogram).  An object owns a copy of the packed reads (0.4 bytes per base) and the lists (4 bytes per
 * base): keep it between the two stages while memory allows, else free it after the tally and let the coverage stage
 * partition again (lrb_packed_cov_hist_many).  in_workspace != 0: the buffers are the context's workspaces instead --
 * nothing is allocated (a 16 GB hipMalloc costs 0.4 s, forty times the partition pass it would save a one-shot run) and
 * the lists are valid until the next call that uses those workspaces (lrb_winlists_valid; a stale object is refused);
 * such an object holds no copy of the reads' codes (it reaches them in the batches' own buffers): the batches must
 * outlive it.
 * lrb_packed_k15_accumulate_half: one batch, one atomic per window. */
typedef struct lrb_winlists lrb_winlists;
int lrb_packed_lists_create(lrb_ctx *ctx, const lrb_packed *const *packs, uint64_t count, int bins, int in_workspace,
                            lrb_winlists **out);
int lrb_winlists_valid(const lrb_ctx *ctx, const lrb_winlists *w, int *valid);
int lrb_winlists_info(const lrb_winlists *w, uint64_t *n_reads, uint64_t *device_bytes, uint32_t *reads_per_group);
int lrb_winlists_tally(lrb_ctx *ctx, const lrb_winlists *w, uint32_t *d_half);
int lrb_winlists_cov_hist(lrb_ctx *ctx, const lrb_winlists *w, const uint8_t *d_map, int bins);
int lrb_winlists_free(lrb_ctx *ctx, lrb_winlists *w);
int lrb_packed_k15_accumulate_half(lrb_ctx *ctx, const lrb_packed *p, uint32_t *d_half);
/* K2 of many resident batches as the product runs it when it does not keep the lists: consecutive batches in groups of
 * at most 4e9 bases, each group cut into window lists in the context's workspaces and tallied into the canonical half;
 * a group below LRB_K2_LISTS_MIN_BASES (33 M) bases by one atomic a window.  count-15mers.cpp:97-123's job. */
int lrb_packed_k15_tally_half_many(lrb_ctx *ctx, const lrb_packed *const *packs, uint64_t count, uint32_t *d_half);
/* ... for a coverage histogram of `bins` bins to follow (1..256; the plain call means 32: lists cut for 32 bins hold any
 * narrower histogram).  The groups are filled FROM THE END (lrb_packed_group_starts), and the LAST group's lists -- with
 * the bounds and the laid-out masks -- are left standing in the workspaces: lrb_packed_cov_hist_many of exactly those
 * batches then sweeps them as they are instead of partitioning the windows a second time (kmer_utils.h:24-87 reads the
 * table count-15mers.cpp:97-123 wrote; nothing is allocated for it).  They stand until a call uses those workspaces
 * (another partition, the text formatter, lrb_ctx_trim) or one of the batches is freed; lrb_packed_lists_resident asks.
 * LRB_RESIDENT_LISTS=0 turns the hand-over off. */
int lrb_packed_k15_tally_half_many_for(lrb_ctx *ctx, const lrb_packed *const *packs, uint64_t count, uint32_t *d_half,
                                       int bins);
int lrb_packed_lists_resident(const lrb_ctx *ctx, const lrb_packed *const *packs, uint64_t count, int bins, int *yes);
/* starts[g] = first batch of group g of `count` consecutive batches cut into groups of at most max_bases bases, filled
 * from the end; starts[*n_groups] = count (the caller gives count + 1 entries). */
int lrb_packed_group_starts(const lrb_packed *const *packs, uint64_t count, uint64_t max_bases, uint64_t *starts,
                            uint64_t *n_groups);


/* ---- K4: clustering distances ----------------------------------------- */
/* calc_distances (cluster_utils.py:45-49): d_out[i] = 0.5 - <M[i], M[seed]>,
 * d_out[seed] = 0.  M is row-major float32 [n_rows x dims], dims <= 64. */
int lrb_seed_dist_dev(lrb_ctx *ctx, const float *d_M, uint64_t n_rows, int dims,
                      uint64_t seed, float *d_out);
/* The histogram step of get_cluster_center (cluster_utils.py:137-139,175-177)
 * for many seeds in one pass over M: d_hist[s*60 + b] = torch.histc(
 * calc_distances(M, seeds[s]), 60, 0, 0.3)[b] as uint32 (the caller does
 * hist[0] -= 1). */
int lrb_seed_hist_dev(lrb_ctx *ctx, const float *d_M, uint64_t n_rows, int dims,
                      const int64_t *d_seeds, uint32_t n_seeds, uint32_t *d_hist);

/* ---- K5: left-over read assignment -------------------------------------- */
/* normal() + the argmax loop of perform_binning (cluster_utils.py:261-268,309-322):
 * d_best[u] = first cluster c maximising sum_f log(N(X[u,f]; mean[c,f], std[c,f]) + 1e-7)
 * in float64, -1 when every cluster evaluates to nan (a zero std gives nan and never
 * wins).  d_best_p (optional) receives the winning value.  Row-major inputs. */
int lrb_gauss_assign_dev(lrb_ctx *ctx, const double *d_X, uint64_t n_rows, int feats,
                         const double *d_mean, const double *d_std, int n_clusters,
                         int32_t *d_best, double *d_best_p);

/* ---- K6: HDBSCAN for the contigs pipeline -------------------------------- */
/* The distance work of hdbscan.HDBSCAN(min_cluster_size=250).fit_predict(latent)
 * (perform_contig_binning_HDBSCAN, cluster_utils.py:483-495) -- a third-party package the
 * reference does not pin; what is restated is the published HDBSCAN* algorithm with that
 * package's defaults (euclidean, alpha 1, excess of mass, no single cluster).
 *
 * d_core[i] = euclidean distance from row i of d_X (row-major n x dims float32,
 * dims <= 64) to its k-th nearest row, the row itself included (k = min_samples); exact
 * (radix select over all n squared distances). */
int lrb_hdb_core_dist_dev(lrb_ctx *ctx, const float *d_X, uint64_t n, int dims, uint32_t k,
                          float *d_core);
/* Minimum spanning tree of the complete graph under the mutual reachability distance
 * max(core[a], core[b], |a - b|) (Boruvka; ties by (weight, lower index, higher index)).
 * Synchronous; writes the n-1 edges to the HOST arrays h_u, h_v (endpoints), h_w (weight);
 * *rounds (optional) = Boruvka rounds taken. */
int lrb_hdb_mst_dev(lrb_ctx *ctx, const float *d_X, uint64_t n, int dims, const float *d_core,
                    uint32_t *h_u, uint32_t *h_v, float *h_w, uint32_t *rounds);
/* Host only: spanning tree -> single linkage -> condensed tree (min_cluster_size) ->
 * stabilities -> excess-of-mass selection -> labels[n] (-1 = noise, clusters numbered from 0
 * in order of appearance in the condensed tree). */
int lrb_hdb_labels(uint64_t n, const uint32_t *u, const uint32_t *v, const float *w,
                   uint32_t min_cluster_size, int32_t *labels, uint32_t *n_clusters);
/* All of the above for a host matrix: labels = HDBSCAN(min_cluster_size, min_samples).  min_samples is cut to
 * n - 1 as the package does (fewer points than min_samples is not an error: nothing can reach min_cluster_size, all
 * noise).  core_excludes_self picks the neighbour the core distance is taken to: 0 = the min_samples-th with the point
 * itself counted (sklearn.cluster.HDBSCAN; the package's Prim's paths), 1 = the min_samples-th OTHER point (the
 * package's Boruvka paths, which algorithm='best' takes for the reference's euclidean latents), -1 = the library
 * default: 1, or 0 with LRB_HDB_CORE=self in the environment.  lrb_hdbscan_host = lrb_hdbscan_host_ex(..., -1, ...). */
int lrb_hdbscan_host_ex(lrb_ctx *ctx, const float *X, uint64_t n, int dims, uint32_t min_cluster_size,
                        uint32_t min_samples, int core_excludes_self, int32_t *labels, uint32_t *n_clusters);
int lrb_hdbscan_host(lrb_ctx *ctx, const float *X, uint64_t n, int dims, uint32_t min_cluster_size,
                     uint32_t min_samples, int32_t *labels, uint32_t *n_clusters);

/* Host only: CPython's random.shuffle on an int64 array, drawing from the Mersenne Twister state
 * the caller took from random.getstate() (mt[624], *pos) and puts back with random.setstate --
 * the shuffle of all remaining read ids before every cluster (cluster_utils.py:219-221), same
 * stream as the interpreter's, 100x faster. */
int lrb_mt_shuffle_i64(uint32_t *mt, int *pos, int64_t *x, uint64_t n);

/* ---- K7: fused VAE training step ----------------------------------------- */
/* VAE.forward + calc_loss + backward + Adam of ae_utils.py:163-241,243-271 as ~20 fused
 * fp32 kernels per step, recorded in a hipGraph.  Architecture as in
 * VAE.__init__ (ae_utils.py:35-97): blocks BatchNorm(Dropout(LeakyReLU(Linear))) over
 * hidden[0..n_hidden), heads mu / softplus(logsigma), mirrored decoder, linear output.
 * loss_weights = {e_cov_weight, e_comp_weight, kld_weight} of hyper_params.json.
 *
 * Parameter vector layout (float32; what the set/get calls move):
 *   per encoder block: W[N][K], b[N], bn.weight[N], bn.bias[N];  then [mu.W; logsigma.W]
 *   ([2L][K]), [mu.b; logsigma.b];  per decoder block the same four;  output W[D][K], b[D].
 * Running statistics vector: per BatchNorm (encoder blocks, then decoder blocks)
 * running_mean[N], running_var[N]. */
typedef struct lrb_vae lrb_vae;
int lrb_vae_create(lrb_ctx *ctx, int cov_size, int prof_size, const int *hidden, int n_hidden,
                   int latent, int max_batch, const float *loss_weights, float lr, float dropout,
                   uint64_t seed, lrb_vae **out);
int lrb_vae_destroy(lrb_vae *v);
int lrb_vae_sizes(lrb_vae *v, uint64_t *n_params, uint64_t *n_running);
/* what: 0 parameters, 1 running statistics, 2 Adam m, 3 Adam v, 4 loss sums {loss, e_cov,
 * e_comp, kld} accumulated over the steps since they were last set. */
int lrb_vae_set(lrb_vae *v, int what, const float *host, uint64_t count);
int lrb_vae_get(lrb_vae *v, int what, float *host, uint64_t count);
int lrb_vae_steps_done(lrb_vae *v, uint64_t *steps);
/* n_steps optimisation steps (trainepoch, ae_utils.py:199-241): step s trains on the rows
 * d_perm[s*batch_size .. (s+1)*batch_size) of d_data (row-major n_rows x (cov+prof)
 * float32, already scaled).  Enqueued on the context's stream; use_graph = 1 records the
 * step once per batch size and replays it. */
int lrb_vae_train_dev(lrb_vae *v, const float *d_data, const int64_t *d_perm, uint32_t batch_size,
                      uint32_t n_steps, int use_graph);
/* VAE.encode (ae_utils.py:141-161): eval mode (running statistics, no dropout), output = mu:
 * d_mu[n_rows][latent] float32 in input order, for the parameters currently in the object. */
int lrb_vae_encode_dev(lrb_vae *v, const float *d_data, uint64_t n_rows, float *d_mu);
/* Test hook: internal buffers of the last step (0 eps, 1 z, 2 mu|logsigma, 3 dL/drecon,
 * 10+i / 20+i encoder / decoder block outputs, 30 the per-slice parameter gradients). */
int lrb_vae_debug_read(lrb_vae *v, int which, float *host, uint64_t count);

/* ---- host side: ingest and profile text -------------------------------- */
/* FASTA/FASTQ(.gz) reader with the record semantics of SeqReader::get_seq
 * (io_utils.h:133-165) over kseq_read (kseq.h:177-218). */
typedef struct lrb_reader lrb_reader;
int lrb_reader_open(const char *path, lrb_reader **out);
/* Next batch of at most max_reads records / about max_bytes bases.  *seqs and
 * *offs stay valid until the next call on this reader.  *n == 0 at the end. */
int lrb_reader_next(lrb_reader *rd, uint64_t max_reads, uint64_t max_bytes,
                    const uint8_t **seqs, const uint64_t **offs, uint64_t *n);
int lrb_reader_close(lrb_reader *rd);

/* The records of a contigs FASTA(.gz) as Bio.SeqIO.parse(path, "fasta") yields them (pipelines.py:125-131): a line
 * starting with '>' opens a record, id = header up to the first white space, sequence = the other lines stripped of
 * surrounding white space and joined.  The whole file is held in host memory: record r is
 * seqs[offs[r] .. offs[r+1]) with id names[name_offs[r] .. name_offs[r+1]).  lrb_fasta_write_fragments writes
 * split_contigs' fragments file (runners_utils.py:53-75: records of >= 5000 bases become windows of 2500 plus the
 * last 2500; ">{record}_{fragment}" ids) and reports how many fragments each record gave. */
typedef struct lrb_fasta_records lrb_fasta_records;
int lrb_fasta_scan(const char *path, lrb_fasta_records **out);
int lrb_fasta_records_view(const lrb_fasta_records *r, uint64_t *n, const uint8_t **seqs, const uint64_t **offs,
                           const uint8_t **names, const uint64_t **name_offs);
int lrb_fasta_write_fragments(const lrb_fasta_records *r, const char *out_path, uint64_t *n_fragments,
                              uint32_t *frags_per_record);
int lrb_fasta_records_free(lrb_fasta_records *r);

/* The same records from a pool of parser threads (plain FASTA is cut into byte ranges of
 * about chunk_bytes, one batch per range, handed out in file order; gzip and FASTQ input
 * run on the serial reader behind the same calls).  LRB_ERR_FORMAT from _next means the
 * file cannot be cut (a '+' line inside FASTA): start over with lrb_reader_*. */
typedef struct lrb_preader lrb_preader;
int lrb_preader_open(const char *path, int threads, uint64_t chunk_bytes, lrb_preader **out);
/* One shard of the same ranges: only ranges rank, rank+world, ... are parsed and handed out
 * (multi-GPU ingest: every rank opens the file with its own rank; the ranges of all ranks
 * tile the file).  A file that falls back to the serial reader is NOT sharded -- check
 * lrb_preader_info. */
int lrb_preader_open_shard(const char *path, int threads, uint64_t chunk_bytes, uint32_t rank,
                           uint32_t world, lrb_preader **out);
int lrb_preader_next(lrb_preader *rd, const uint8_t **seqs, const uint64_t **offs, uint64_t *n);
/* The same reader whose pool also PACKS every batch into the HBM layout (codes 2 bits a base + validity mask, the
 * regions of lrb_pack_layout) in the thread that parsed it: flags = LRB_PREADER_PACKED.  After lrb_preader_next,
 * lrb_preader_packed_view hands out the packed arrays of that batch (library memory, valid until the next _next) for
 * lrb_packed_create_packed -- 0.375 bytes a base cross PCIe instead of 1.  A file on the serial reader (gzip, FASTQ) has
 * no packed view (LRB_ERR_ARG): pack such a batch with lrb_pack_reads_host (sizes from lrb_pack_host_sizes). */
#define LRB_PREADER_PACKED 1u
int lrb_preader_open_ex(const char *path, int threads, uint64_t chunk_bytes, uint32_t rank, uint32_t world, uint32_t flags,
                        lrb_preader **out);
int lrb_preader_packed_view(lrb_preader *rd, const uint32_t **codes, const uint32_t **mask, const uint64_t **code_off,
                            const uint64_t **mask_off, const uint32_t **lens);
int lrb_pack_host_sizes(const uint64_t *offs, uint64_t n, uint64_t *code_words, uint64_t *mask_words);
int lrb_pack_reads_host(const uint8_t *seqs, const uint64_t *offs, uint64_t n, uint32_t *codes, uint32_t *mask,
                        uint64_t *code_off, uint64_t *mask_off, uint32_t *lens);
int lrb_pack_reads_host_scalar(const uint8_t *seqs, const uint64_t *offs, uint64_t n, uint32_t *codes, uint32_t *mask,
                               uint64_t *code_off, uint64_t *mask_off, uint32_t *lens); /* the same without AVX2 / BMI2 (tests) */
/* *parallel = 1 if the pool is parsing byte ranges (0: serial reader behind the calls);
 * *n_ranges = ranges in the whole file; *last_range = index of the range the last _next
 * returned.  Any pointer may be NULL. */
int lrb_preader_info(lrb_preader *rd, int *parallel, uint64_t *n_ranges, uint64_t *last_range);
int lrb_preader_close(lrb_preader *rd);

/* com_profs rows (count-kmers.cpp:89-92,110-118): value = count/max(1,len-k+1)
 * printed "%f" + ' ' after every value, '\n' per read.  vals (optional,
 * n*dim doubles) receives the 6-decimal values the text holds, i.e. what
 * pipelines.py:315-321 later parses into com_profs.npy.  buf needs
 * lrb_profile_text_bound(n, dim) bytes. */
uint64_t lrb_profile_text_bound(uint64_t n, uint32_t dim);
int lrb_format_com(const uint32_t *counts, const uint32_t *lens, uint64_t n, uint32_t dim,
                   int k, int threads, char *buf, uint64_t *written, double *vals);
/* cov_profs rows (kmer_utils.h:74-84, search-15mers.cpp:35-47): hist/sum,
 * < 1e-4 -> 0, "%f" separated by single spaces, no trailing space. */
int lrb_format_cov(const uint32_t *hist, const uint32_t *sums, uint64_t n, uint32_t bins,
                   int threads, char *buf, uint64_t *written, double *vals);

/* ---- K8: profile rows as text, on the device ------------------------------ */
/* Replaces the serial to_string loops of count-kmers.cpp:110-118 and search-15mers.cpp:35-47
 * (and lrb_format_com / lrb_format_cov above for counts that are already in HBM).  Every value is
 * a ratio in [0, 1], so "%f" is always 8 characters and a row has a fixed width:
 * lrb_com_row_bytes(dim) = 9 * dim + 1 (a space after every value, '\n'), lrb_cov_row_bytes(bins)
 * = 9 * bins (single spaces between, '\n').  d_text: n rows of that width, 16-byte aligned;
 * d_q6 (optional): u32 [n][dim], the six-decimal integer of every value.  Rounding is glibc's
 * (exact binary value, ties to even).  A value above 1 (counts that exceed their row total) is
 * LRB_ERR_ARG -- format such input with the host functions.  Synchronises the stream. */
uint64_t lrb_com_row_bytes(uint32_t dim);
uint64_t lrb_cov_row_bytes(uint32_t bins);
int lrb_format_com_dev(lrb_ctx *ctx, const uint32_t *d_counts, const uint32_t *d_lens, uint64_t n,
                       uint32_t dim, int k, uint8_t *d_text, uint32_t *d_q6);
int lrb_format_cov_dev(lrb_ctx *ctx, const uint32_t *d_hist, const uint32_t *d_sums, uint64_t n,
                       uint32_t bins, uint8_t *d_text, uint32_t *d_q6);

/* test hook: the library's "%f" and libc's for one value (64-byte buffers) */
int lrb_debug_format_f(double v, char *ours, char *libc);

#ifdef __cplusplus
}
#endif
#endif /* LRB_HIP_H */
