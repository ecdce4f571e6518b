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
#include <zlib.h>

#include <algorithm>
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
// profile text
// ---------------------------------------------------------------------------
extern "C" uint64_t lrb_profile_text_bound(uint64_t n, uint32_t dim)
{
    // values are in [0,1]: "d.dddddd" + one separator, plus the newline
    return n * ((uint64_t)dim * 10 + 2) + 16;
}

namespace {

// "%f" of v (0 <= v <= 1e9) written at p; returns the byte count and the value the
// text holds (digits / 1e6, correctly rounded == what float(token) parses).
inline int put_f(char *p, double v, double *parsed)
{
    const int len = snprintf(p, 32, "%f", v);
    if (parsed) {
        // strip the decimal point: the token is <int>.<6 digits>
        uint64_t q = 0;
        for (int i = 0; i < len; ++i)
            if (p[i] != '.') q = q * 10 + (uint64_t)(p[i] - '0');
        *parsed = (double)q / 1e6;
    }
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
