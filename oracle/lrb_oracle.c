/*
 * lrb_oracle.c -- CPU restatement of LRBinner's profile hot path (TEST INFRASTRUCTURE).
 *
 * This file is the parity oracle for the HIP path in lrbinner_amd/csrc.  It is
 * NOT part of the product: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it, and only as the checker.
 *
 * Every function cites the reference code (under /root/reference, never
 * copied) whose behaviour it restates.  The restatement is pinned against the
 * real reference binaries (oracle/_ref, built by oracle/Makefile) by
 * tests/test_oracle_vs_ref.py and against the committed fixtures in
 * tests/golden/ (generated from those binaries by tests/golden/make_golden.py).
 *
 * Plain C99, no dependencies besides libc + zlib (reader only).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

/* ------------------------------------------------------------------ */
/* 2-bit base code: (ascii >> 1) & 3  =>  A=0 C=1 T=2 G=3.             */
/* count-kmers.cpp:77, kmer_utils.h:47,131                              */
static inline uint32_t base_code(uint8_t c) { return (uint32_t)((c >> 1) & 3u); }

/* Strict validity used by the 15-mer paths only: uppercase ACGT.       */
/* kmer_utils.h:38-43,122-127                                           */
static inline int base_valid(uint8_t c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }

/* Reverse complement of a k-mer packed 2 bits/base (first base in the  */
/* most significant of the 2k bits).  Reverse the 2-bit groups of the   */
/* 64-bit word, complement (XOR 10b per group), shift down.             */
/* count-kmers.cpp:24-36, kmer_utils.h:10-22                            */
uint64_t orc_revcomp(uint64_t x, unsigned k)
{
    uint64_t r = 0;
    for (unsigned i = 0; i < k; i++) {
        uint64_t g = (x >> (2 * i)) & 3u; /* base k-1-i of the k-mer   */
        r = (r << 2) | (g ^ 2u);          /* complement: A<->T, C<->G  */
    }
    return r;
}

/* Canonical-index LUT: walk codes 0..4^k-1 ascending; a code whose     */
/* reverse complement was already numbered shares that number, else it  */
/* takes the next free one.  Returns D (32/136/512 for k=3/4/5).        */
/* count-kmers.cpp:38-64                                                */
uint32_t orc_kmer_lut(unsigned k, uint32_t *lut)
{
    const uint64_t n = 1ull << (2 * k);
    uint32_t next = 0;
    for (uint64_t c = 0; c < n; c++) {
        uint64_t rc = orc_revcomp(c, k);
        if (rc < c)
            lut[c] = lut[rc];
        else
            lut[c] = next++;
    }
    return next;
}

/* Integer view of count_kmers: every byte is coded (no validity test,  */
/* no reset); from the k-th byte on each window tallies LUT[val].       */
/* total = number of windows = max(0, len-k+1).                         */
/* count-kmers.cpp:66-87                                                */
void orc_count_kmers(const uint8_t *seq, uint64_t len, unsigned k, const uint32_t *lut,
                     uint32_t D, uint32_t *counts, uint64_t *total)
{
    const uint64_t mask = (1ull << (2 * k)) - 1;
    uint64_t val = 0, t = 0;
    memset(counts, 0, sizeof(uint32_t) * D);
    for (uint64_t i = 0; i < len; i++) {
        val = ((val << 2) & mask) + base_code(seq[i]);
        if (i + 1 >= k) {
            counts[lut[val]]++;
            t++;
        }
    }
    if (total) *total = t;
}

/* profile[i] = count[i] / max(1.0, total) in double.                   */
/* count-kmers.cpp:89-92                                                */
void orc_com_profile(const uint32_t *counts, uint32_t D, uint64_t total, double *prof)
{
    double den = (double)total;
    if (den < 1.0) den = 1.0;
    for (uint32_t i = 0; i < D; i++) prof[i] = (double)counts[i] / den;
}

/* Batch driver over concatenated reads (offs has n+1 entries).         */
void orc_count_kmers_batch(const uint8_t *seqs, const uint64_t *offs, uint64_t n, unsigned k,
                           uint32_t *counts /* n x D */, uint64_t *totals /* n or NULL */)
{
    uint32_t *lut = (uint32_t *)malloc(sizeof(uint32_t) << (2 * k));
    uint32_t D = orc_kmer_lut(k, lut);
    for (uint64_t r = 0; r < n; r++) {
        uint64_t t;
        orc_count_kmers(seqs + offs[r], offs[r + 1] - offs[r], k, lut, D, counts + r * D, &t);
        if (totals) totals[r] = t;
    }
    free(lut);
}

/* ------------------------------------------------------------------ */
/* 15-mers.  A window is usable only if its 15 bytes are all uppercase  */
/* ACGT; any other byte resets the window.  kmer_utils.h:114-137         */
#define K15_MASK 1073741823ull /* 4^15 - 1 */

/* Emit the forward code of every valid 15-mer of one read, in order.   */
/* Returns the number emitted (<= len-14).                              */
uint64_t orc_k15_emit(const uint8_t *seq, uint64_t len, uint32_t *out)
{
    uint64_t val = 0, n = 0;
    unsigned run = 0;
    for (uint64_t i = 0; i < len; i++) {
        if (!base_valid(seq[i])) {
            val = 0;
            run = 0;
            continue;
        }
        val = ((val << 2) & K15_MASK) + base_code(seq[i]);
        if (run < 15) run++;
        if (run == 15) out[n++] = (uint32_t)val;
    }
    return n;
}

/* Dense accumulate: T[val]++ and T[rc(val)]++ (uint32 wrap) for every  */
/* valid 15-mer.  kmer_utils.h:139-153                                  */
void orc_k15_accumulate_dense(const uint8_t *seq, uint64_t len, uint32_t *table)
{
    uint64_t val = 0;
    unsigned run = 0;
    for (uint64_t i = 0; i < len; i++) {
        if (!base_valid(seq[i])) {
            val = 0;
            run = 0;
            continue;
        }
        val = ((val << 2) & K15_MASK) + base_code(seq[i]);
        if (run < 15) run++;
        if (run == 15) {
            table[val]++;
            table[orc_revcomp(val, 15)]++;
        }
    }
}

static int cmp_u32(const void *a, const void *b)
{
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return (x > y) - (x < y);
}

/* Sparse form of the same table: sorted unique slot indices + their    */
/* counts (every slot not listed is 0).  keys/counts must hold          */
/* 2 * (number of valid 15-mers) entries.  Returns #unique.             */
uint64_t orc_k15_sparse(const uint8_t *seqs, const uint64_t *offs, uint64_t n,
                        uint32_t *keys, uint32_t *counts)
{
    uint64_t m = 0;
    for (uint64_t r = 0; r < n; r++) {
        uint64_t len = offs[r + 1] - offs[r];
        uint64_t got = orc_k15_emit(seqs + offs[r], len, keys + m);
        m += got;
    }
    /* second half: the reverse complements */
    for (uint64_t i = 0; i < m; i++) keys[m + i] = (uint32_t)orc_revcomp(keys[i], 15);
    m *= 2;
    qsort(keys, m, sizeof(uint32_t), cmp_u32);
    uint64_t u = 0;
    for (uint64_t i = 0; i < m;) {
        uint64_t j = i;
        while (j < m && keys[j] == keys[i]) j++;
        keys[u] = keys[i];
        counts[u] = (uint32_t)(j - i); /* wraps like the uint32 table */
        u++;
        i = j;
    }
    return u;
}

/* Upper bound on valid 15-mers of a batch (for sizing keys/counts).    */
uint64_t orc_k15_max_windows(const uint64_t *offs, uint64_t n)
{
    uint64_t m = 0;
    for (uint64_t r = 0; r < n; r++) {
        uint64_t len = offs[r + 1] - offs[r];
        if (len >= 15) m += len - 14;
    }
    return m;
}

static uint32_t sparse_lookup(const uint32_t *keys, const uint32_t *counts, uint64_t nu, uint32_t key)
{
    uint64_t lo = 0, hi = nu;
    while (lo < hi) {
        uint64_t mid = (lo + hi) >> 1;
        if (keys[mid] < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    return (lo < nu && keys[lo] == key) ? counts[lo] : 0u;
}

/* The coverage bin of one table count.  kmer_utils.h:54-69             */
/*   count < 2 -> 0;  pos = count / bin_size - 1 (integer)              */
/*   count <= bin_size          -> bin 0                                */
/*   0 < pos < bins             -> bin pos                              */
/*   else                       -> bin bins-1                           */
static inline long cov_bin(long count, long bin_size, int bins)
{
    count = count < 2 ? 0 : count;
    long pos = (count / bin_size) - 1;
    if (count <= bin_size) return 0;
    if (pos < bins && pos > 0) return pos;
    return bins - 1;
}

long orc_cov_bin(long count, long bin_size, int bins) { return cov_bin(count, bin_size, bins); }

/* Integer view of line_to_vec: per valid 15-mer (forward code only)    */
/* look the count up, bin it, tally.  sum = number of valid 15-mers.    */
/* table_dense may be NULL, then (keys,counts,nu) is the sparse table.  */
/* kmer_utils.h:24-72                                                   */
void orc_cov_hist(const uint8_t *seq, uint64_t len, const uint32_t *table_dense,
                  const uint32_t *keys, const uint32_t *counts, uint64_t nu,
                  long bin_size, int bins, uint32_t *hist, uint64_t *sum)
{
    uint64_t val = 0, s = 0;
    unsigned run = 0;
    memset(hist, 0, sizeof(uint32_t) * (size_t)bins);
    for (uint64_t i = 0; i < len; i++) {
        if (!base_valid(seq[i])) {
            val = 0;
            run = 0;
            continue;
        }
        val = ((val << 2) & K15_MASK) + base_code(seq[i]);
        if (run < 15) run++;
        if (run == 15) {
            long c = table_dense ? (long)table_dense[val]
                                 : (long)sparse_lookup(keys, counts, nu, (uint32_t)val);
            hist[cov_bin(c, bin_size, bins)]++;
            s++;
        }
    }
    if (sum) *sum = s;
}

void orc_cov_hist_batch(const uint8_t *seqs, const uint64_t *offs, uint64_t n,
                        const uint32_t *table_dense, const uint32_t *keys,
                        const uint32_t *counts, uint64_t nu, long bin_size, int bins,
                        uint32_t *hist /* n x bins */, uint64_t *sums /* n */)
{
    for (uint64_t r = 0; r < n; r++)
        orc_cov_hist(seqs + offs[r], offs[r + 1] - offs[r], table_dense, keys, counts, nu,
                     bin_size, bins, hist + r * (uint64_t)bins, sums ? sums + r : NULL);
}

/* counts[i] /= sum; values < 1e-4 -> 0; sum == 0 leaves zeros.         */
/* kmer_utils.h:74-84                                                   */
void orc_cov_profile(const uint32_t *hist, int bins, uint64_t sum, double *prof)
{
    for (int i = 0; i < bins; i++) {
        double v = (double)hist[i];
        if (sum > 0) {
            v /= (double)(long)sum;
            if (v < 1e-4) v = 0;
        }
        prof[i] = v;
    }
}

/* ------------------------------------------------------------------ */
/* Text rows.  std::to_string(double) == "%f".                          */
/* com_profs: every value followed by ' ', then '\n'                    */
/*   (count-kmers.cpp:110-118)                                          */
/* cov_profs: values separated by single ' ', no trailing space, '\n'   */
/*   (search-15mers.cpp:35-47)                                          */
/* Returns bytes written (buf must hold 25*(n+1) bytes).                */
uint64_t orc_format_com_row(const double *v, uint32_t n, char *buf)
{
    char *p = buf;
    for (uint32_t i = 0; i < n; i++) {
        p += sprintf(p, "%f", v[i]);
        *p++ = ' ';
    }
    *p++ = '\n';
    return (uint64_t)(p - buf);
}

uint64_t orc_format_cov_row(const double *v, uint32_t n, char *buf)
{
    char *p = buf;
    for (uint32_t i = 0; i < n; i++) {
        p += sprintf(p, "%f", v[i]);
        if (i + 1 < n) *p++ = ' ';
    }
    *p++ = '\n';
    return (uint64_t)(p - buf);
}

/* ------------------------------------------------------------------ */
/* FASTA/FASTQ reader with the record semantics of kseq_read as driven  */
/* by SeqReader::get_seq (io_utils.h:133-165, kseq.h:177-218):          */
/*  - skip to the first '>' or '@' anywhere in the stream               */
/*  - name = up to first whitespace; rest of header line is a comment   */
/*  - sequence lines are joined; a line whose FIRST byte is '>', '@' or */
/*    '+' ends the sequence; empty lines are skipped; a trailing '\r'   */
/*    on a line is dropped when the joined sequence is longer than 1    */
/*  - '+' starts a FASTQ quality block: skip that line, then read       */
/*    quality lines until at least as many bytes as the sequence; a     */
/*    missing or length-mismatched quality block ends the stream        */
/*  - the sequence is handed on as a C string (cut at the first NUL)    */
/* ------------------------------------------------------------------ */
typedef struct {
    gzFile f;
    unsigned char buf[16384];
    int beg, end, eof;
} orc_stream;

static int st_getc(orc_stream *s)
{
    if (s->eof && s->beg >= s->end) return -1;
    if (s->beg >= s->end) {
        s->beg = 0;
        s->end = gzread(s->f, s->buf, sizeof s->buf);
        if (s->end <= 0) {
            s->eof = 1;
            s->end = 0;
            return -1;
        }
    }
    return (int)s->buf[s->beg++];
}

typedef struct {
    char *s;
    size_t l, m;
} orc_str;

static void str_push(orc_str *t, int c)
{
    if (t->l + 2 > t->m) {
        t->m = t->m ? t->m * 2 : 256;
        t->s = (char *)realloc(t->s, t->m);
    }
    t->s[t->l++] = (char)c;
}

/* Append the rest of the current line (without '\n'); applies the      */
/* trailing-'\r' rule of ks_getuntil2.  Returns -1 at EOF with nothing  */
/* read, else 0.                                                        */
static int st_restofline(orc_stream *s, orc_str *t)
{
    int c, got = 0;
    /* kseq reports "nothing" only when the stream is already exhausted */
    if (s->eof && s->beg >= s->end) return -1;
    while ((c = st_getc(s)) >= 0) {
        got = 1;
        if (c == '\n') break;
        str_push(t, c);
    }
    if (!got && c < 0) return -1;
    if (t->l > 1 && t->s[t->l - 1] == '\r') t->l--;
    return 0;
}

typedef struct {
    uint8_t *seqs;
    uint64_t *offs;
    uint64_t n, cap_n, bytes, cap_b;
} orc_reads;

static void reads_push(orc_reads *R, const char *s, size_t l)
{
    if (R->n + 2 > R->cap_n) {
        R->cap_n = R->cap_n ? R->cap_n * 2 : 1024;
        R->offs = (uint64_t *)realloc(R->offs, sizeof(uint64_t) * (R->cap_n + 1));
    }
    if (R->bytes + l + 1 > R->cap_b) {
        while (R->bytes + l + 1 > R->cap_b) R->cap_b = R->cap_b ? R->cap_b * 2 : (1u << 20);
        R->seqs = (uint8_t *)realloc(R->seqs, R->cap_b);
    }
    if (R->n == 0) R->offs[0] = 0;
    memcpy(R->seqs + R->bytes, s, l);
    R->bytes += l;
    R->n++;
    R->offs[R->n] = R->bytes;
}

/* Reads the whole file.  The seqs and offs arrays are malloc'd (free     */
/* with orc_free).  Returns 0, or -1 if the file cannot be opened.       */
int orc_fastx_read(const char *path, uint8_t **seqs, uint64_t **offs, uint64_t *n)
{
    orc_stream *s = (orc_stream *)calloc(1, sizeof *s);
    orc_reads R;
    orc_str seq = {0, 0, 0}, qual = {0, 0, 0};
    int c, last = 0;
    memset(&R, 0, sizeof R);
    s->f = gzopen(path, "r");
    if (!s->f) {
        free(s);
        return -1;
    }
    for (;;) {
        if (last == 0) {
            while ((c = st_getc(s)) >= 0 && c != '>' && c != '@') {}
            if (c < 0) break;
            last = c;
        }
        /* header: name up to whitespace, then the rest of the line */
        {
            int got = 0;
            if (s->eof && s->beg >= s->end) break;
            while ((c = st_getc(s)) >= 0) {
                got = 1;
                if (c == ' ' || (c >= '\t' && c <= '\r')) break;
            }
            if (!got && c < 0) break; /* header char was the last byte */
            if (c >= 0 && c != '\n') {
                while ((c = st_getc(s)) >= 0 && c != '\n') {}
            }
        }
        seq.l = 0;
        while ((c = st_getc(s)) >= 0 && c != '>' && c != '+' && c != '@') {
            if (c == '\n') continue;
            str_push(&seq, c);
            st_restofline(s, &seq);
        }
        if (c == '>' || c == '@') last = c;
        str_push(&seq, 0);
        seq.l--;
        if (c != '+') {
            reads_push(&R, seq.s, strlen(seq.s));
            if (c < 0) break;
            continue;
        }
        /* FASTQ */
        while ((c = st_getc(s)) >= 0 && c != '\n') {}
        if (c < 0) break; /* no quality string: stream ends, record dropped */
        qual.l = 0;
        while (st_restofline(s, &qual) >= 0 && qual.l < seq.l) {}
        last = 0;
        if (qual.l != seq.l) break; /* mismatch: stream ends, record dropped */
        reads_push(&R, seq.s, strlen(seq.s));
    }
    gzclose(s->f);
    free(s);
    free(seq.s);
    free(qual.s);
    if (R.n == 0) {
        R.offs = (uint64_t *)calloc(1, sizeof(uint64_t));
        R.seqs = (uint8_t *)malloc(1);
    }
    *seqs = R.seqs;
    *offs = R.offs;
    *n = R.n;
    return 0;
}

void orc_free(void *p) { free(p); }

/* ------------------------------------------------------------------ */
/* Synthetic data for the accuracy gate (no reference counterpart: the  */
/* Sim-8 set of README.md:70-76 is an external download).  A genome is  */
/* an order-`order` Markov chain over ACGT; cum[ctx*4 + b] is the        */
/* cumulative transition table (4^order contexts).  The random stream   */
/* is splitmix64, so the bytes are the same on every platform.          */
static inline uint64_t splitmix64(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void orc_synth_markov(uint64_t seed, unsigned order, const double *cum, uint64_t len, uint8_t *out)
{
    static const char L[4] = { 'A', 'C', 'G', 'T' };
    uint64_t s = seed, nctx = 1ull << (2 * order), ctx = 0;
    for (uint64_t i = 0; i < len; i++) {
        double u = (double)(splitmix64(&s) >> 11) * (1.0 / 9007199254740992.0);
        const double *c = cum + 4 * ctx;
        unsigned b = (u >= c[0]) + (u >= c[1]) + (u >= c[2]);
        out[i] = (uint8_t)L[b];
        ctx = ((ctx << 2) | b) & (nctx - 1);
    }
}

/* One noisy read: `src[0..n)` copied to `dst` with substitutions (p_sub), deletions */
/* (p_del) and insertions (p_ins) per base, then reverse-complemented when `rc`.    */
/* Returns the length written (dst must hold 2*n).                                  */
uint64_t orc_synth_read(uint64_t seed, const uint8_t *src, uint64_t n, double p_sub, double p_del,
                        double p_ins, int rc, uint8_t *dst)
{
    static const char L[4] = { 'A', 'C', 'G', 'T' };
    uint64_t s = seed, m = 0;
    for (uint64_t i = 0; i < n && m + 2 < 2 * n; i++) {
        double u = (double)(splitmix64(&s) >> 11) * (1.0 / 9007199254740992.0);
        if (u < p_del)
            continue;
        if (u < p_del + p_sub)
            dst[m++] = (uint8_t)L[splitmix64(&s) & 3];
        else
            dst[m++] = src[i];
        if (u > 1.0 - p_ins)
            dst[m++] = (uint8_t)L[splitmix64(&s) & 3];
    }
    if (rc) {
        for (uint64_t i = 0, j = m; i < j--; i++) {
            uint8_t a = dst[i], b = dst[j];
            dst[i] = b; dst[j] = a;
        }
        for (uint64_t i = 0; i < m; i++) {
            uint8_t c = dst[i];
            dst[i] = c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : 'C';
        }
    }
    return m;
}
