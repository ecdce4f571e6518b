// lrb_host.cpp -- host half of the C ABI (include/lrb_hip.h): sequence ingest and
// the text form of the profiles.  No GPU code here.
//
//   lrb_reader_*   FASTA/FASTQ(.gz) records with the semantics of SeqReader::get_seq
//                  (io_utils.h:133-165) driving kseq_read (kseq.h:177-218)
//   lrb_format_com com_profs rows  (count-kmers.cpp:89-92,110-118)
//   lrb_format_cov cov_profs rows  (kmer_utils.h:74-84, search-15mers.cpp:35-47)
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <zlib.h>

#include <algorithm>
#include <new>
#include <thread>
#include <vector>

#include "lrb_hip.h"
#include "lrb_internal.h"

// ---------------------------------------------------------------------------
// reader
// ---------------------------------------------------------------------------
namespace {

constexpr size_t kChunk = 4u << 20;

struct ByteStream {
    gzFile f = nullptr;
    std::vector<uint8_t> buf;
    size_t beg = 0, end = 0;
    bool eof = false;

    bool refill()
    {
        if (eof) return false;
        int got = gzread(f, buf.data(), (unsigned)buf.size());
        beg = 0;
        if (got <= 0) {
            end = 0;
            eof = true;
            return false;
        }
        end = (size_t)got;
        return true;
    }
    bool exhausted() const { return eof && beg >= end; }
    int getc()
    {
        if (beg >= end && !refill()) return -1;
        return buf[beg++];
    }
    // Append bytes up to (not including) the next '\n' to dst and consume the '\n'.
    // Returns false when the stream was already exhausted (nothing consumed).
    bool take_line(std::vector<uint8_t> &dst)
    {
        if (exhausted()) return false;
        bool any = false;
        for (;;) {
            if (beg >= end && !refill()) break;
            any = true;
            const uint8_t *s = buf.data() + beg;
            const uint8_t *nl = (const uint8_t *)memchr(s, '\n', end - beg);
            const size_t take = nl ? (size_t)(nl - s) : end - beg;
            dst.insert(dst.end(), s, s + take);
            beg += take + (nl ? 1 : 0);
            if (nl) break;
        }
        return any;
    }
    // Skip through the next '\n'.  Returns false if the stream ends first.
    bool skip_line()
    {
        for (;;) {
            if (beg >= end && !refill()) return false;
            const uint8_t *s = buf.data() + beg;
            const uint8_t *nl = (const uint8_t *)memchr(s, '\n', end - beg);
            if (nl) {
                beg += (size_t)(nl - s) + 1;
                return true;
            }
            beg = end;
        }
    }
};

inline bool is_space(int c) { return c == ' ' || (c >= '\t' && c <= '\r'); }

} // namespace

struct lrb_reader {
    ByteStream in;
    int pending = 0; // header character already consumed ('>' or '@'), 0 = none
    bool done = false;
    std::vector<uint8_t> seqs; // current batch, records back to back
    std::vector<uint64_t> offs;
    std::vector<uint8_t> qual;
};

// One record appended to rd->seqs.  Returns false at the end of the stream.
static bool next_record(lrb_reader *rd)
{
    ByteStream &in = rd->in;
    int c;
    if (rd->pending == 0) {
        // hunt for the first header character, anywhere in the stream
        while ((c = in.getc()) >= 0 && c != '>' && c != '@') {}
        if (c < 0) return false;
        rd->pending = c;
    }
    // header line: the name ends at the first white space; anything after it up to
    // the end of the line is a comment.  A header character that is the very last
    // byte of the stream yields no record.
    if (in.exhausted()) return false;
    bool any = false;
    while ((c = in.getc()) >= 0) {
        any = true;
        if (is_space(c)) break;
    }
    if (!any) return false;
    if (c >= 0 && c != '\n') in.skip_line();

    std::vector<uint8_t> &S = rd->seqs;
    const size_t start = S.size();
    // sequence lines until a line that STARTS with '>', '@' or '+'
    while ((c = in.getc()) >= 0 && c != '>' && c != '+' && c != '@') {
        if (c == '\n') continue;
        S.push_back((uint8_t)c);
        if (in.take_line(S)) {
            // a trailing CR goes when the joined sequence is longer than one byte
            if (S.size() - start > 1 && S.back() == '\r') S.pop_back();
        }
    }
    if (c == '>' || c == '@') rd->pending = c;
    const size_t raw_len = S.size() - start;

    bool keep = true, more = true;
    if (c == '+') {
        // FASTQ: drop the '+' line, then quality lines until they cover the sequence
        if (!in.skip_line()) {
            keep = false; // no quality block: the stream ends here
            more = false;
        } else {
            std::vector<uint8_t> &Q = rd->qual;
            Q.clear();
            for (;;) {
                if (!in.take_line(Q)) break;
                if (Q.size() > 1 && Q.back() == '\r') Q.pop_back();
                if (Q.size() >= raw_len) break;
            }
            rd->pending = 0;
            if (Q.size() != raw_len) {
                keep = false; // length mismatch ends the stream
                more = false;
            }
        }
    } else if (c < 0) {
        more = false;
    }
    if (!keep) {
        S.resize(start);
        rd->done = true;
        return false;
    }
    // handed on as a C string: cut at the first NUL byte
    if (raw_len) {
        const uint8_t *z = (const uint8_t *)memchr(S.data() + start, 0, raw_len);
        if (z) S.resize((size_t)(z - S.data()));
    }
    rd->offs.push_back(S.size());
    if (!more) rd->done = true;
    return true;
}

extern "C" int lrb_reader_open(const char *path, lrb_reader **out)
{
    if (!path || !out) {
        lrb_set_error("invalid argument: %s%s", "path/out is null", "");
        return LRB_ERR_ARG;
    }
    gzFile f = gzopen(path, "rb");
    if (!f) {
        lrb_set_error("cannot open %s%s", path, "");
        return LRB_ERR_IO;
    }
    gzbuffer(f, 1u << 20);
    lrb_reader *rd = new (std::nothrow) lrb_reader();
    if (!rd) {
        gzclose(f);
        return LRB_ERR_NOMEM;
    }
    rd->in.f = f;
    rd->in.buf.resize(kChunk);
    *out = rd;
    return LRB_OK;
}

extern "C" int lrb_reader_next(lrb_reader *rd, uint64_t max_reads, uint64_t max_bytes,
                               const uint8_t **seqs, const uint64_t **offs, uint64_t *n)
{
    if (!rd || !seqs || !offs || !n) {
        lrb_set_error("invalid argument: %s%s", "null pointer", "");
        return LRB_ERR_ARG;
    }
    rd->seqs.clear();
    rd->offs.clear();
    rd->offs.push_back(0);
    try {
        while (!rd->done && rd->offs.size() - 1 < max_reads && rd->seqs.size() < max_bytes) {
            if (!next_record(rd)) {
                rd->done = true;
                break;
            }
        }
        rd->seqs.reserve(rd->seqs.size() + 64); // readable slack for staged copies
    } catch (const std::bad_alloc &) {
        lrb_set_error("out of host memory while reading%s%s", "", "");
        return LRB_ERR_NOMEM;
    }
    if (rd->seqs.empty()) rd->seqs.reserve(64);
    *seqs = rd->seqs.data();
    *offs = rd->offs.data();
    *n = rd->offs.size() - 1;
    return LRB_OK;
}

extern "C" int lrb_reader_close(lrb_reader *rd)
{
    if (!rd) return LRB_OK;
    if (rd->in.f) gzclose(rd->in.f);
    delete rd;
    return LRB_OK;
}

// ---------------------------------------------------------------------------
// Contigs: the records of a FASTA file as the reference's contigs pipeline sees them
// (Bio.SeqIO.parse(path, "fasta"): pipelines.py:125-131, runners_utils.py:53-75, cluster_utils.py:512-530) --
// a line that STARTS with '>' opens a record, its id is the header up to the first white space, the sequence is
// the other lines with their TRAILING white space stripped, joined, and every ' ' and '\r' taken out wherever it
// stands (Bio's SimpleFastaParser: `"".join(lines).replace(" ", "").replace("\r", "")` over `line.rstrip()`s --
// blank-separated blocks of ten and stray carriage returns vanish, a leading tab stays); lines before the first
// header are ignored; '@' and '+' mean nothing here.  One pass in native code: a contigs file wrapped at 60 columns is
// fifty million lines, which is half a minute of a Python loop.
// ---------------------------------------------------------------------------
struct lrb_fasta_records {
    std::vector<uint8_t> seqs, names;
    std::vector<uint64_t> offs, name_offs; // n + 1 each
};

namespace {
inline bool py_space(uint8_t c) { return c == ' ' || (c >= '\t' && c <= '\r'); } // bytes.strip() / bytes.split()
// take ' ' and '\r' out of v[from..) in place (memchr first: almost no line has any)
inline void drop_blank_cr(std::vector<uint8_t> &v, size_t from)
{
    const size_t n = v.size() - from;
    if (n == 0 || (!memchr(v.data() + from, ' ', n) && !memchr(v.data() + from, '\r', n))) return;
    size_t w = from;
    for (size_t i = from; i < v.size(); ++i)
        if (v[i] != ' ' && v[i] != '\r') v[w++] = v[i];
    v.resize(w);
}
}

extern "C" int lrb_fasta_scan(const char *path, lrb_fasta_records **out)
{
    if (!path || !out) {
        lrb_set_error("invalid argument: %s%s", "path/out is null", "");
        return LRB_ERR_ARG;
    }
    gzFile f = gzopen(path, "rb");
    if (!f) {
        lrb_set_error("cannot open %s%s", path, "");
        return LRB_ERR_IO;
    }
    gzbuffer(f, 1u << 20);
    lrb_fasta_records *r = new (std::nothrow) lrb_fasta_records;
    if (!r) {
        gzclose(f);
        return LRB_ERR_NOMEM;
    }
    ByteStream in;
    in.f = f;
    in.buf.resize(kChunk);
    std::vector<uint8_t> line;
    bool open = false;
    try {
        // a plain file's sequences take no more room than the file: one allocation instead of a doubling vector
        // (whose reallocations copy every byte again and fault every page twice)
        if (gzdirect(f)) {
            FILE *probe = fopen(path, "rb");
            if (probe) {
                if (fseeko(probe, 0, SEEK_END) == 0 && ftello(probe) > 0) {
                    r->seqs.reserve((size_t)ftello(probe));
                    // gigabytes of fresh pages: ask for huge ones (first touch is most of this function's time)
                    const uintptr_t a = ((uintptr_t)r->seqs.data() + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1);
                    const uintptr_t e = ((uintptr_t)r->seqs.data() + r->seqs.capacity()) & ~(uintptr_t)((2u << 20) - 1);
                    if (e > a) (void)madvise((void *)a, e - a, MADV_HUGEPAGE);
                }
                fclose(probe);
            }
        }
        r->offs.push_back(0);
        r->name_offs.push_back(0);
        // one text line as Bio's parser sees it (universal newlines: a lone '\r' ends a line too, so `piece` holds no '\r')
        auto piece = [&](const uint8_t *p, size_t n) {
            if (n && p[0] == '>') {
                if (open) r->offs.push_back(r->seqs.size());
                open = true;
                size_t a = 1;
                while (a < n && py_space(p[a])) ++a;
                size_t b = a;
                while (b < n && !py_space(p[b])) ++b;
                r->names.insert(r->names.end(), p + a, p + b);
                r->name_offs.push_back(r->names.size());
            } else if (open) {
                size_t b = n;
                while (b > 0 && py_space(p[b - 1])) --b;
                const size_t from = r->seqs.size();
                r->seqs.insert(r->seqs.end(), p, p + b);
                drop_blank_cr(r->seqs, from);
            }
        };
        // ... and a line of the file: pieces between '\r's (a "\r\n" ending leaves an empty last piece: nothing)
        auto text_line = [&](const std::vector<uint8_t> &l) {
            size_t a = 0;
            for (;;) {
                const uint8_t *cr = a < l.size() ? (const uint8_t *)memchr(l.data() + a, '\r', l.size() - a) : nullptr;
                const size_t e = cr ? (size_t)(cr - l.data()) : l.size();
                piece(l.data() + a, e - a);
                if (!cr) break;
                a = e + 1;
            }
        };
        for (;;) {
            // a sequence line goes straight into the record (one copy); only header lines, lines that start with white
            // space and lines with a '\r' inside take the detour through `line`
            if (open && (in.beg < in.end || in.refill())) {
                const uint8_t c0 = in.buf[in.beg];
                if (c0 != '>' && !py_space(c0)) {
                    const size_t from = r->seqs.size();
                    in.take_line(r->seqs);
                    // (a '\r' as the last byte is a "\r\n" ending: stripped below like any trailing white space)
                    const size_t n = r->seqs.size() - from;
                    const uint8_t *cr = n > 1 ? (const uint8_t *)memchr(r->seqs.data() + from, '\r', n - 1) : nullptr;
                    if (cr) {
                        line.assign(r->seqs.begin() + from, r->seqs.end());
                        r->seqs.resize(from);
                        text_line(line);
                        continue;
                    }
                    while (r->seqs.size() > from && py_space(r->seqs.back())) r->seqs.pop_back();
                    drop_blank_cr(r->seqs, from);
                    continue;
                }
            }
            line.clear();
            if (!in.take_line(line)) break;
            text_line(line);
        }
        if (open) r->offs.push_back(r->seqs.size());
    } catch (const std::bad_alloc &) {
        gzclose(f);
        delete r;
        lrb_set_error("out of memory while reading %s%s", path, "");
        return LRB_ERR_NOMEM;
    }
    gzclose(f);
    *out = r;
    return LRB_OK;
}

extern "C" int lrb_fasta_records_view(const lrb_fasta_records *r, uint64_t *n, const uint8_t **seqs, const uint64_t **offs,
                                      const uint8_t **names, const uint64_t **name_offs)
{
    if (!r) {
        lrb_set_error("invalid argument: %s%s", "records is null", "");
        return LRB_ERR_ARG;
    }
    if (n) *n = r->offs.size() - 1;
    if (seqs) *seqs = r->seqs.data();
    if (offs) *offs = r->offs.data();
    if (names) *names = r->names.data();
    if (name_offs) *name_offs = r->name_offs.data();
    return LRB_OK;
}

// split_contigs (runners_utils.py:53-75): a record of >= 5000 bases becomes windows of 2500 plus its last 2500
// bases, a shorter one stays whole; fragment i of record n is written as ">{n}_{i}\n{bases}\n", i counting through
// the whole file.  frags_per_record[n] receives the number of fragments of record n.
extern "C" int lrb_fasta_write_fragments(const lrb_fasta_records *r, const char *out_path, uint64_t *n_fragments,
                                         uint32_t *frags_per_record)
{
    if (!r || !out_path) {
        lrb_set_error("invalid argument: %s%s", "records/out_path is null", "");
        return LRB_ERR_ARG;
    }
    FILE *f = fopen(out_path, "wb");
    if (!f) {
        lrb_set_error("cannot open %s for writing%s", out_path, "");
        return LRB_ERR_IO;
    }
    std::vector<char> buf(8u << 20);
    setvbuf(f, buf.data(), _IOFBF, buf.size());
    const uint64_t n = r->offs.size() - 1;
    uint64_t i = 0;
    bool ok = true;
    char head[64];
    auto put = [&](uint64_t rec, const uint8_t *p, uint64_t len) {
        const int h = snprintf(head, sizeof head, ">%llu_%llu\n", (unsigned long long)rec, (unsigned long long)i);
        ok = ok && fwrite(head, 1, (size_t)h, f) == (size_t)h && fwrite(p, 1, len, f) == len && fputc('\n', f) != EOF;
        ++i;
    };
    for (uint64_t rec = 0; rec < n && ok; ++rec) {
        const uint8_t *s = r->seqs.data() + r->offs[rec];
        const uint64_t len = r->offs[rec + 1] - r->offs[rec];
        const uint64_t before = i;
        if (len >= 5000) {
            for (uint64_t x = 0; x < len; x += 2500) put(rec, s + x, len - x < 2500 ? len - x : 2500);
            put(rec, s + len - 2500, 2500);
        } else {
            put(rec, s, len);
        }
        if (frags_per_record) frags_per_record[rec] = (uint32_t)(i - before);
    }
    if (fclose(f) != 0) ok = false;
    if (!ok) {
        lrb_set_error("write to %s failed%s", out_path, "");
        return LRB_ERR_IO;
    }
    if (n_fragments) *n_fragments = i;
    return LRB_OK;
}

extern "C" int lrb_fasta_records_free(lrb_fasta_records *r)
{
    delete r;
    return LRB_OK;
}

// ---------------------------------------------------------------------------
// parallel reader: plain (uncompressed) FASTA parsed by a pool of threads.
//
// The file is mapped and cut into byte ranges.  The worker of range i owns every
// record whose header line STARTS inside the range; a header is a line whose first
// byte is '>' or '@' (exactly the lines that end a sequence in kseq_read), except for
// the very first record of the file, which kseq finds at the first '>' or '@' anywhere.
// Each range becomes one batch, handed out in file order.  gzip input, FASTQ input
// (first header '@') and LRB_SERIAL_READER=1 use the serial reader behind the same
// calls.  A line starting with '+' inside FASTA (a quality block, whose length rule
// could swallow later headers) cannot be cut into independent ranges: the reader
// reports LRB_ERR_FORMAT and the caller starts over with the serial reader.
// ---------------------------------------------------------------------------
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <condition_variable>
#include <map>
#include <mutex>

#include <immintrin.h>

// ---------------------------------------------------------------------------
// The packed HBM layout made on the HOST (round 5): what pack_kernel writes from ASCII on the device -- codes 2 bits a
// base, first base in bits 31..30, EVERY byte coded (c >> 1) & 3; mask 1 bit a base, first base in bit 31, 1 iff the
// byte is one of ACGT; regions of roundup4(ceil(L/16)) + 4 and roundup4(ceil(L/32)) + 4 words, zero padded
// (lrb_pack_layout) -- so that 0.375 bytes a base cross PCIe instead of 1.  The parser pool does it per range, in the
// thread that parsed the range.  32 bases a step: four pext (bits 2..1 of eight byte-swapped bytes each) for the codes,
// four byte compares + a byte-reversed movemask for the validity bits; a scalar loop for machines without AVX2 / BMI2
// and for a read's last partial block.
// ---------------------------------------------------------------------------
static inline uint64_t host_code_words(uint64_t L) { return ((((L + 15) >> 4) + 3) & ~3ull) + 4; }
static inline uint64_t host_mask_words(uint64_t L) { return ((((L + 31) >> 5) + 3) & ~3ull) + 4; }

static inline void pack_block_scalar(const uint8_t *p, uint32_t nb, uint32_t *c0, uint32_t *c1, uint32_t *m)
{
    uint32_t a = 0, b = 0, v = 0;
    for (uint32_t i = 0; i < nb; ++i) {
        const uint32_t ch = p[i], code = (ch >> 1) & 3u;
        if (i < 16) a |= code << (30 - 2 * i);
        else b |= code << (30 - 2 * (i - 16));
        v |= (uint32_t)((ch == 'A') | (ch == 'C') | (ch == 'G') | (ch == 'T')) << (31 - i);
    }
    *c0 = a;
    *c1 = b;
    *m = v;
}

__attribute__((target("avx2,bmi2"))) static void pack_read_avx2(const uint8_t *p, uint64_t L, uint32_t *cw, uint32_t *mw)
{
    const __m256i rev = _mm256_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
    const __m256i A = _mm256_set1_epi8('A'), C = _mm256_set1_epi8('C'), G = _mm256_set1_epi8('G'), T = _mm256_set1_epi8('T');
    const uint64_t full = L >> 5;
    for (uint64_t c = 0; c < full; ++c) {
        const uint8_t *q = p + (c << 5);
        uint64_t x[4];
        memcpy(x, q, 32);
        const uint32_t h0 = (uint32_t)_pext_u64(__builtin_bswap64(x[0]), 0x0606060606060606ull);
        const uint32_t h1 = (uint32_t)_pext_u64(__builtin_bswap64(x[1]), 0x0606060606060606ull);
        const uint32_t h2 = (uint32_t)_pext_u64(__builtin_bswap64(x[2]), 0x0606060606060606ull);
        const uint32_t h3 = (uint32_t)_pext_u64(__builtin_bswap64(x[3]), 0x0606060606060606ull);
        cw[2 * c] = (h0 << 16) | h1;
        cw[2 * c + 1] = (h2 << 16) | h3;
        const __m256i v = _mm256_loadu_si256((const __m256i *)q);
        const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(v, A), _mm256_cmpeq_epi8(v, C)),
                                           _mm256_or_si256(_mm256_cmpeq_epi8(v, G), _mm256_cmpeq_epi8(v, T)));
        // byte 31 - i of the shuffled vector = byte i: movemask then has base 0 in bit 31
        const __m256i r = _mm256_permute2x128_si256(_mm256_shuffle_epi8(ok, rev), _mm256_shuffle_epi8(ok, rev), 0x01);
        mw[c] = (uint32_t)_mm256_movemask_epi8(r);
    }
    if (L & 31u) pack_block_scalar(p + (full << 5), (uint32_t)(L & 31u), &cw[2 * full], &cw[2 * full + 1], &mw[full]);
}

static void pack_read_scalar(const uint8_t *p, uint64_t L, uint32_t *cw, uint32_t *mw)
{
    const uint64_t full = L >> 5;
    for (uint64_t c = 0; c < full; ++c) pack_block_scalar(p + (c << 5), 32, &cw[2 * c], &cw[2 * c + 1], &mw[c]);
    if (L & 31u) pack_block_scalar(p + (full << 5), (uint32_t)(L & 31u), &cw[2 * full], &cw[2 * full + 1], &mw[full]);
}

// codes / mask / offsets / lengths of n reads (offs: byte offsets into seqs, offs[0] need not be 0), laid out from word 0
static int pack_reads_host_impl(const uint8_t *seqs, const uint64_t *offs, uint64_t n, uint32_t *codes, uint32_t *mask,
                                uint64_t *code_off, uint64_t *mask_off, uint32_t *lens, bool allow_simd);

extern "C" int lrb_pack_reads_host(const uint8_t *seqs, const uint64_t *offs, uint64_t n, uint32_t *codes, uint32_t *mask,
                                   uint64_t *code_off, uint64_t *mask_off, uint32_t *lens)
{
    return pack_reads_host_impl(seqs, offs, n, codes, mask, code_off, mask_off, lens, true);
}

// (tests: the scalar loop, whatever the machine has)
extern "C" int lrb_pack_reads_host_scalar(const uint8_t *seqs, const uint64_t *offs, uint64_t n, uint32_t *codes, uint32_t *mask,
                                          uint64_t *code_off, uint64_t *mask_off, uint32_t *lens)
{
    return pack_reads_host_impl(seqs, offs, n, codes, mask, code_off, mask_off, lens, false);
}

static int pack_reads_host_impl(const uint8_t *seqs, const uint64_t *offs, uint64_t n, uint32_t *codes, uint32_t *mask,
                                uint64_t *code_off, uint64_t *mask_off, uint32_t *lens, bool allow_simd)
{
    if (n && (!seqs || !offs || !codes || !mask || !code_off || !mask_off)) {
        lrb_set_error("invalid argument: %s%s", "null pointer", "");
        return LRB_ERR_ARG;
    }
    static const bool have = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2");
    const bool fast = have && allow_simd;
    uint64_t co = 0, mo = 0;
    for (uint64_t r = 0; r < n; ++r) {
        if (offs[r + 1] < offs[r] || offs[r + 1] - offs[r] >= 0xFFFFFFFFull) {
            lrb_set_error("read too long for the packed layout (>= 2^32-1 bases)%s%s", "", "");
            return LRB_ERR_ARG;
        }
        const uint64_t L = offs[r + 1] - offs[r], ncw = host_code_words(L), nmw = host_mask_words(L);
        code_off[r] = co;
        mask_off[r] = mo;
        if (lens) lens[r] = (uint32_t)L;
        uint32_t *cw = codes + co, *mw = mask + mo;
        const uint64_t blocks = (L + 31) >> 5; // words written by the packers: 2 blocks of codes, blocks of mask
        if (fast) pack_read_avx2(seqs + offs[r], L, cw, mw);
        else pack_read_scalar(seqs + offs[r], L, cw, mw);
        memset(cw + 2 * blocks, 0, (ncw - 2 * blocks) * 4);
        memset(mw + blocks, 0, (nmw - blocks) * 4);
        co += ncw;
        mo += nmw;
    }
    if (code_off) code_off[n] = co;
    if (mask_off) mask_off[n] = mo;
    return LRB_OK;
}

extern "C" int lrb_pack_host_sizes(const uint64_t *offs, uint64_t n, uint64_t *code_words, uint64_t *mask_words)
{
    if (n && !offs) {
        lrb_set_error("invalid argument: %s%s", "null offsets", "");
        return LRB_ERR_ARG;
    }
    uint64_t co = 0, mo = 0;
    for (uint64_t r = 0; r < n; ++r) {
        const uint64_t L = offs[r + 1] - offs[r];
        co += host_code_words(L);
        mo += host_mask_words(L);
    }
    if (code_words) *code_words = co;
    if (mask_words) *mask_words = mo;
    return LRB_OK;
}

namespace {

struct PBatch {
    std::vector<uint8_t> seqs;
    std::vector<uint64_t> offs;
    // the same batch packed on the host (readers opened with LRB_PREADER_PACKED)
    std::vector<uint32_t> codes, mask, lens;
    std::vector<uint64_t> code_off, mask_off;
    bool bad = false;        // '+' line met
    bool pack_failed = false; // the host packer refused the batch (a read of 2^32 - 1 bases or more)
    bool packed_ok = false;   // codes / mask / offsets / lens describe THIS batch (not what a recycled buffer held)
};

// first header at or after `from`: a '>' / '@' that begins a line
inline size_t next_header(const uint8_t *d, size_t size, size_t from)
{
    size_t p = from;
    if (p == 0) {
        if (size && (d[0] == '>' || d[0] == '@')) return 0;
    } else if (d[p - 1] == '\n' && p < size && (d[p] == '>' || d[p] == '@')) {
        return p;
    }
    while (p < size) {
        const uint8_t *nl = (const uint8_t *)memchr(d + p, '\n', size - p);
        if (!nl) return size;
        p = (size_t)(nl - d) + 1;
        if (p < size && (d[p] == '>' || d[p] == '@')) return p;
    }
    return size;
}

// Parse the records whose header starts in [start, limit).  `start` points at a header
// character.  Same joining / CR / empty-line rules as next_record() above.
void parse_range(const uint8_t *d, size_t size, size_t start, size_t limit, PBatch *out)
{
    std::vector<uint8_t> &S = out->seqs;
    out->offs.push_back(0);
    size_t p = start;
    while (p < size && p < limit) {
        // header line: skipped whole (name and comment are not used by the path)
        ++p; // the header character
        if (p >= size) break; // a header character that is the last byte yields no record
        {
            const uint8_t *nl = (const uint8_t *)memchr(d + p, '\n', size - p);
            p = nl ? (size_t)(nl - d) + 1 : size;
        }
        const size_t rec0 = S.size();
        // sequence lines until a line that starts with a header character
        while (p < size) {
            const uint8_t c = d[p];
            if (c == '>' || c == '@') break;
            if (c == '+') {
                out->bad = true;
                return;
            }
            if (c == '\n') {
                ++p;
                continue;
            }
            const uint8_t *nl = (const uint8_t *)memchr(d + p, '\n', size - p);
            const size_t e = nl ? (size_t)(nl - d) : size;
            S.insert(S.end(), d + p, d + e);
            // kseq appends the first byte, then the rest of the line; the CR rule applies
            // only when a "rest of line" read happened, i.e. unless that first byte was the
            // last byte of the stream
            const bool had_rest = !(e == size && e - p == 1);
            if (had_rest && S.size() - rec0 > 1 && S.back() == '\r') S.pop_back();
            p = nl ? e + 1 : size;
        }
        const size_t raw = S.size() - rec0;
        if (raw) {
            const uint8_t *z = (const uint8_t *)memchr(S.data() + rec0, 0, raw);
            if (z) S.resize((size_t)(z - S.data()));
        }
        out->offs.push_back(S.size());
    }
    if (S.capacity() < S.size() + 64) S.reserve(S.size() + 64);
}

} // namespace

struct lrb_preader {
    // serial fallback
    lrb_reader *serial = nullptr;
    uint64_t chunk_bytes = 0;
    // parallel mode
    int fd = -1;
    const uint8_t *data = nullptr;
    size_t size = 0;
    size_t first = 0; // first header of the file
    size_t n_chunks = 0, next_issue = 0, next_take = 0; // next_* count THIS shard's ranges
    size_t shard_rank = 0, shard_world = 1, n_own = 0, last_index = 0;
    int max_ahead = 0;
    std::vector<std::thread> pool;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::map<size_t, PBatch *> done;
    std::vector<PBatch *> spare; // recycled batches: their pages are already faulted in
    bool stop = false;
    bool packed = false; // every batch is also packed into the HBM layout by the thread that parsed it
    PBatch *current = nullptr;

    PBatch *fresh()
    {
        PBatch *b = nullptr;
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!spare.empty()) {
                b = spare.back();
                spare.pop_back();
            }
        }
        if (!b) {
            b = new PBatch();
            b->seqs.reserve(chunk_bytes + (1u << 20));
            b->offs.reserve(chunk_bytes / 2000 + 1024);
        }
        b->seqs.clear();
        b->offs.clear();
        b->bad = false;
        b->pack_failed = false;
        b->packed_ok = false;
        return b;
    }
    void recycle(PBatch *b)
    {
        if (!b) return;
        std::lock_guard<std::mutex> lk(mu);
        spare.push_back(b);
    }

    void worker()
    {
        for (;;) {
            size_t i;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || (next_issue < n_own && next_issue < next_take + (size_t)max_ahead); });
                if (stop) return;
                i = next_issue++;
            }
            PBatch *b = fresh();
            const size_t ci = shard_rank + i * shard_world; // byte range of the file
            const size_t lo = ci * chunk_bytes, hi = std::min(size, (ci + 1) * (size_t)chunk_bytes);
            // the file's first record starts at `first` wherever that is (kseq hunts for the
            // first header character anywhere); every later record starts at a line start
            size_t start;
            if (first >= lo && first < hi)
                start = first;
            else if (lo > first)
                start = next_header(data, size, lo);
            else
                start = size; // a range that ends before the first header
            if (start < hi) parse_range(data, size, start, hi, b);
            else b->offs.push_back(0);
            if (packed && !b->bad && b->offs.size() > 1) {
                const uint64_t nr = b->offs.size() - 1;
                uint64_t cwn = 0, mwn = 0;
                lrb_pack_host_sizes(b->offs.data(), nr, &cwn, &mwn);
                b->codes.resize(cwn);      // (recycled batches keep their capacity: no page faults after the first round)
                b->mask.resize(mwn);
                b->lens.resize(nr);
                b->code_off.resize(nr + 1);
                b->mask_off.resize(nr + 1);
                if (lrb_pack_reads_host(b->seqs.data(), b->offs.data(), nr, b->codes.data(), b->mask.data(), b->code_off.data(),
                                        b->mask_off.data(), b->lens.data()) != LRB_OK)
                    b->pack_failed = true;
                else
                    b->packed_ok = true;
            }
            {
                // This range will not be read again: drop its page-table entries now, here, in parallel, so
                // that closing the reader does not end in one multi-second munmap of the whole file (which
                // holds the address-space lock against every page fault of the process).  A neighbour that
                // still runs a few bytes into the range (its last record) simply faults them back in.
                const size_t page = 4096, a = (lo + page - 1) & ~(page - 1), z = hi & ~(page - 1);
                if (z > a) madvise((void *)(data + a), z - a, MADV_DONTNEED);
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                done[i] = b;
            }
            cv_done.notify_all();
        }
    }
};

extern "C" int lrb_preader_open(const char *path, int threads, uint64_t chunk_bytes, lrb_preader **out)
{
    return lrb_preader_open_shard(path, threads, chunk_bytes, 0, 1, out);
}

extern "C" int lrb_preader_open_shard(const char *path, int threads, uint64_t chunk_bytes, uint32_t rank,
                                      uint32_t world, lrb_preader **out)
{
    return lrb_preader_open_ex(path, threads, chunk_bytes, rank, world, 0, out);
}

extern "C" int lrb_preader_open_ex(const char *path, int threads, uint64_t chunk_bytes, uint32_t rank, uint32_t world,
                                   uint32_t flags, lrb_preader **out)
{
    if (!path || !out || world < 1 || rank >= world) {
        lrb_set_error("invalid argument: %s%s", "path/out is null", "");
        return LRB_ERR_ARG;
    }
    if (threads < 1) threads = 1;
    if (chunk_bytes < 64) chunk_bytes = 64;
    lrb_preader *pr = new (std::nothrow) lrb_preader();
    if (!pr) return LRB_ERR_NOMEM;
    pr->chunk_bytes = chunk_bytes;
    bool serial = getenv("LRB_SERIAL_READER") != nullptr && getenv("LRB_SERIAL_READER")[0] == '1';
    int fd = open(path, O_RDONLY);
    if (fd < 0) {
        delete pr;
        lrb_set_error("cannot open %s%s", path, "");
        return LRB_ERR_IO;
    }
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) serial = true;
    const size_t size = serial ? 0 : (size_t)st.st_size;
    const uint8_t *data = nullptr;
    if (!serial && size > 0) {
        void *m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) serial = true;
        else data = (const uint8_t *)m;
    }
    if (!serial && size >= 2 && data[0] == 0x1f && data[1] == 0x8b) serial = true; // gzip
    size_t first = 0;
    if (!serial) {
        while (first < size && data[first] != '>' && data[first] != '@') ++first;
        if (first < size && data[first] == '@') serial = true; // FASTQ: quality blocks are sequential
    }
    if (serial) {
        if (data) munmap((void *)data, size);
        close(fd);
        int rc = lrb_reader_open(path, &pr->serial);
        if (rc != LRB_OK) {
            delete pr;
            return rc;
        }
        *out = pr;
        return LRB_OK;
    }
    madvise((void *)data, size, MADV_SEQUENTIAL);
    pr->fd = fd;
    pr->data = data;
    pr->size = size;
    pr->first = first;
    pr->n_chunks = size ? (size + chunk_bytes - 1) / chunk_bytes : 0;
    pr->shard_rank = rank;
    pr->shard_world = world;
    pr->n_own = pr->n_chunks > rank ? (pr->n_chunks - rank + world - 1) / world : 0;
    pr->max_ahead = threads + 2;
    pr->packed = (flags & LRB_PREADER_PACKED) != 0;
    for (int t = 0; t < threads; ++t) pr->pool.emplace_back([pr] { pr->worker(); });
    *out = pr;
    return LRB_OK;
}

extern "C" int lrb_preader_next(lrb_preader *pr, const uint8_t **seqs, const uint64_t **offs,
                                uint64_t *n)
{
    if (!pr || !seqs || !offs || !n) {
        lrb_set_error("invalid argument: %s%s", "null pointer", "");
        return LRB_ERR_ARG;
    }
    if (pr->serial) return lrb_reader_next(pr->serial, ~0ull, pr->chunk_bytes, seqs, offs, n);
    pr->recycle(pr->current);
    pr->current = nullptr;
    for (;;) {
        PBatch *b = nullptr;
        {
            std::unique_lock<std::mutex> lk(pr->mu);
            if (pr->next_take >= pr->n_own) {
                *n = 0;
                static const uint64_t zero = 0;
                static const uint8_t none = 0;
                *seqs = &none;
                *offs = &zero;
                return LRB_OK;
            }
            const size_t want = pr->next_take;
            pr->cv_done.wait(lk, [&] { return pr->done.count(want) != 0; });
            b = pr->done[want];
            pr->done.erase(want);
            pr->next_take++;
            pr->last_index = pr->shard_rank + want * pr->shard_world;
        }
        pr->cv_work.notify_all();
        if (b->bad) {
            delete b;
            lrb_set_error("'+' line inside FASTA: this file needs the serial reader%s%s", "", "");
            return LRB_ERR_FORMAT;
        }
        if (b->pack_failed) {   // its own diagnosis: not a file for the serial reader, a read no batch can hold
            delete b;
            lrb_set_error("a read of 2^32 - 1 bases or more cannot be packed%s%s", "", "");
            return LRB_ERR_ARG;
        }
        if (b->offs.size() <= 1) { // a range without a record start: nothing to hand out
            pr->recycle(b);
            continue;
        }
        pr->current = b;
        *seqs = b->seqs.data();
        *offs = b->offs.data();
        *n = b->offs.size() - 1;
        return LRB_OK;
    }
}

// the batch the last lrb_preader_next returned, in the packed layout (readers opened with LRB_PREADER_PACKED on a file
// the pool parses; LRB_ERR_ARG otherwise -- the serial reader's batches are packed by the caller, lrb_pack_reads_host)
extern "C" int lrb_preader_packed_view(lrb_preader *pr, const uint32_t **codes, const uint32_t **mask, const uint64_t **code_off,
                                       const uint64_t **mask_off, const uint32_t **lens)
{
    if (!pr || pr->serial || !pr->packed || !pr->current || !pr->current->packed_ok) {
        lrb_set_error("invalid argument: %s%s", "no packed batch at hand", "");
        return LRB_ERR_ARG;
    }
    PBatch *b = pr->current;
    if (codes) *codes = b->codes.data();
    if (mask) *mask = b->mask.data();
    if (code_off) *code_off = b->code_off.data();
    if (mask_off) *mask_off = b->mask_off.data();
    if (lens) *lens = b->lens.data();
    return LRB_OK;
}

extern "C" int lrb_preader_info(lrb_preader *pr, int *parallel, uint64_t *n_ranges, uint64_t *last_range)
{
    if (!pr) {
        lrb_set_error("invalid argument: %s%s", "null reader", "");
        return LRB_ERR_ARG;
    }
    if (parallel) *parallel = pr->serial ? 0 : 1;
    if (n_ranges) *n_ranges = pr->serial ? 0 : pr->n_chunks;
    if (last_range) *last_range = pr->serial ? 0 : pr->last_index;
    return LRB_OK;
}

extern "C" int lrb_preader_close(lrb_preader *pr)
{
    if (!pr) return LRB_OK;
    if (pr->serial) {
        lrb_reader_close(pr->serial);
    } else {
        {
            std::lock_guard<std::mutex> lk(pr->mu);
            pr->stop = true;
        }
        pr->cv_work.notify_all();
        for (auto &t : pr->pool) t.join();
        // Tearing down the mapping of a 10 GB file and a few GB of batch buffers is ~0.5 s of page-table
        // work that nobody has to wait for: a detached thread does it while the caller moves on.
        std::vector<PBatch *> junk(pr->spare);
        for (auto &kv : pr->done) junk.push_back(kv.second);
        junk.push_back(pr->current);
        const uint8_t *data = pr->data;
        const size_t size = pr->size;
        const int fd = pr->fd;
        auto teardown = [junk, data, size, fd]() {
            for (auto *b : junk) delete b;
            if (data) munmap((void *)data, size);
            if (fd >= 0) close(fd);
        };
        try {
            std::thread(teardown).detach();
        } catch (...) {
            teardown();
        }
    }
    delete pr;
    return LRB_OK;
}

// ---------------------------------------------------------------------------
// CPython's random.shuffle, same stream
// ---------------------------------------------------------------------------
// cluster_points shuffles every remaining read id before each cluster (cluster_utils.py:
// 219-221, random.shuffle); a run seeded like the reference must draw the same numbers, and
// interpreted that is 0.1 s per 100 k ids.  This is the Mersenne Twister CPython uses
// (MT19937, Matsumoto & Nishimura 1998) with random.py's algorithm on top:
//   shuffle:  for i = n-1 .. 1:  j = randbelow(i + 1);  swap(x[i], x[j])
//   randbelow(m): k = bit_length(m); r = getrandbits(k); while (r >= m) r = getrandbits(k)
//   getrandbits(k <= 32) = genrand_uint32() >> (32 - k)
// mt[624] + *pos are random.getstate()[1]; the caller puts them back with random.setstate.
extern "C" int lrb_mt_shuffle_i64(uint32_t *mt, int *pos, int64_t *x, uint64_t n)
{
    if (!mt || !pos || (!x && n) || *pos < 0 || *pos > 624 || n > 0xFFFFFFFFull) {
        lrb_set_error("invalid argument: %s%s", "lrb_mt_shuffle_i64", "");
        return LRB_ERR_ARG;
    }
    int p = *pos;
    auto next32 = [&]() -> uint32_t {
        if (p >= 624) {
            for (int k = 0; k < 624; ++k) {
                const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7FFFFFFFu);
                mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908B0DFu : 0u);
            }
            p = 0;
        }
        uint32_t y = mt[p++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9D2C5680u;
        y ^= (y << 15) & 0xEFC60000u;
        y ^= y >> 18;
        return y;
    };
    for (uint64_t i = n; i-- > 1;) {
        const uint32_t m = (uint32_t)(i + 1);
        const int k = 32 - __builtin_clz(m);
        uint32_t r = next32() >> (32 - k);
        while (r >= m) r = next32() >> (32 - k);
        const int64_t t = x[i];
        x[i] = x[r];
        x[r] = t;
    }
    *pos = p;
    return LRB_OK;
}

// ---------------------------------------------------------------------------
// profile text
// ---------------------------------------------------------------------------
extern "C" uint64_t lrb_profile_text_bound(uint64_t n, uint32_t dim)
{
    // values are in [0,1]: "d.dddddd" + one separator, plus the newline
    return n * ((uint64_t)dim * 10 + 2) + 16;
}

namespace {

// printf("%f") without printf.  glibc rounds the EXACT binary value of the double to six
// decimals, ties to even; v = m * 2^e with m < 2^53, so m * 10^6 < 2^73 fits 128 bits and
// the rounding can be done on integers: q = round_half_even(m * 10^6 / 2^-e).  Returns the
// byte count; *q6 = the six-decimal value as an integer (value = q6 / 1e6).
// Anything outside 0 <= v < 2^40 (nan, inf, negatives, huge) takes the snprintf path.
inline int put_f_exact(char *p, double v, uint64_t *q6)
{
    uint64_t bits;
    memcpy(&bits, &v, 8);
    const int ex = (int)((bits >> 52) & 0x7FF);
    if ((bits >> 63) || ex >= 1023 + 40) return -1;
    uint64_t m = bits & ((1ull << 52) - 1);
    int e; // v = m * 2^e
    if (ex == 0) {
        e = -1074;
    } else {
        m |= 1ull << 52;
        e = ex - 1075;
    }
    unsigned __int128 x = (unsigned __int128)m * 1000000u;
    uint64_t q;
    if (e >= 0) {
        q = (uint64_t)(x << e);
    } else {
        const int sh = -e;
        if (sh > 127) {
            q = 0; // far below half a unit of the sixth decimal
        } else {
            const unsigned __int128 one = (unsigned __int128)1 << sh;
            const unsigned __int128 rem = x & (one - 1), half = one >> 1;
            q = (uint64_t)(x >> sh);
            if (rem > half || (rem == half && (q & 1))) ++q;
        }
    }
    if (q6) *q6 = q;
    // digits: integer part, '.', six decimals
    const uint64_t ip = q / 1000000u;
    uint32_t fp = (uint32_t)(q % 1000000u);
    char tmp[24];
    int n = 0;
    uint64_t t = ip;
    do {
        tmp[n++] = (char)('0' + t % 10);
        t /= 10;
    } while (t);
    int len = 0;
    while (n) p[len++] = tmp[--n];
    p[len++] = '.';
    for (int i = 5; i >= 0; --i) {
        p[len + i] = (char)('0' + fp % 10);
        fp /= 10;
    }
    return len + 6;
}

// "%f" of v written at p; returns the byte count and the value the text holds
// (digits / 1e6, correctly rounded == what float(token) parses).
inline int put_f(char *p, double v, double *parsed)
{
    uint64_t q = 0;
    int len = put_f_exact(p, v, &q);
    if (len < 0) {
        len = snprintf(p, 32, "%f", v);
        if (parsed) {
            q = 0;
            for (int i = 0; i < len; ++i)
                if (p[i] >= '0' && p[i] <= '9') q = q * 10 + (uint64_t)(p[i] - '0');
        }
    }
    if (parsed) *parsed = (double)q / 1e6;
    return len;
}

template <class RowFn>
int format_rows(uint64_t n, int threads, char *buf, uint64_t *written, RowFn row_fn,
                uint64_t row_bound)
{
    if (threads < 1) threads = 1;
    if ((uint64_t)threads > n) threads = n ? (int)n : 1;
    std::vector<std::vector<char>> parts((size_t)threads);
    std::vector<std::thread> pool;
    bool failed = false;
    auto work = [&](int t) {
        const uint64_t lo = n * (uint64_t)t / threads, hi = n * (uint64_t)(t + 1) / threads;
        try {
            std::vector<char> &out = parts[(size_t)t];
            out.resize((hi - lo) * row_bound + 64);
            char *p = out.data();
            for (uint64_t r = lo; r < hi; ++r) p = row_fn(r, p);
            out.resize((size_t)(p - out.data()));
        } catch (...) {
            failed = true;
        }
    };
    for (int t = 1; t < threads; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto &th : pool) th.join();
    if (failed) {
        lrb_set_error("out of host memory while formatting%s%s", "", "");
        return LRB_ERR_NOMEM;
    }
    uint64_t pos = 0;
    for (auto &part : parts) {
        memcpy(buf + pos, part.data(), part.size());
        pos += part.size();
    }
    *written = pos;
    return LRB_OK;
}

} // namespace

// Self-check hook for the tests: our "%f" next to libc's for one value.
extern "C" int lrb_debug_format_f(double v, char *ours, char *libc)
{
    int n = put_f(ours, v, nullptr);
    ours[n] = 0;
    snprintf(libc, 64, "%f", v);
    return n;
}

extern "C" int lrb_format_com(const uint32_t *counts, const uint32_t *lens, uint64_t n,
                              uint32_t dim, int k, int threads, char *buf, uint64_t *written,
                              double *vals)
{
    if (!buf || !written || (n && (!counts || !lens)) || k < 1) {
        lrb_set_error("invalid argument: %s%s", "lrb_format_com", "");
        return LRB_ERR_ARG;
    }
    auto row = [&](uint64_t r, char *p) -> char * {
        const uint32_t L = lens[r];
        // total = number of windows; the divisor is max(1.0, total)  count-kmers.cpp:91
        const double total = L >= (uint32_t)k ? (double)(L - (uint32_t)k + 1) : 0.0;
        const double den = total < 1.0 ? 1.0 : total;
        const uint32_t *c = counts + r * dim;
        for (uint32_t i = 0; i < dim; ++i) {
            p += put_f(p, (double)c[i] / den, vals ? vals + r * dim + i : nullptr);
            *p++ = ' ';
        }
        *p++ = '\n';
        return p;
    };
    return format_rows(n, threads, buf, written, row, (uint64_t)dim * 10 + 2);
}

extern "C" int lrb_format_cov(const uint32_t *hist, const uint32_t *sums, uint64_t n,
                              uint32_t bins, int threads, char *buf, uint64_t *written,
                              double *vals)
{
    if (!buf || !written || (n && (!hist || !sums)) || bins < 1) {
        lrb_set_error("invalid argument: %s%s", "lrb_format_cov", "");
        return LRB_ERR_ARG;
    }
    auto row = [&](uint64_t r, char *p) -> char * {
        const uint32_t *h = hist + r * bins;
        const double sum = (double)sums[r];
        for (uint32_t i = 0; i < bins; ++i) {
            double v = (double)h[i];
            if (sums[r] > 0) {
                v /= sum;
                if (v < 1e-4) v = 0; // kmer_utils.h:79-82
            }
            p += put_f(p, v, vals ? vals + r * bins + i : nullptr);
            if (i + 1 < bins) *p++ = ' ';
        }
        *p++ = '\n';
        return p;
    };
    return format_rows(n, threads, buf, written, row, (uint64_t)bins * 10 + 2);
}
