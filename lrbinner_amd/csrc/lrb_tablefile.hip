// lrb_tablefile.hip -- the 15-mer table file (kmer_utils.h:89-112) written from and read into device memory:
// whole-table writer (inline and on a job thread), the part writer of the sharded driver (every rank writes its
// slice of ONE file at its final offset), the reader.  No kernels: hipMemcpyAsync through page-locked staging.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <fcntl.h>
#include <unistd.h>

#include <new>
#include <string>
#include <thread>
#include <vector>

#include "lrb_device.h"

// ---- table file ------------------------------------------------------------
// kmer_utils.h:89-112: little-endian u64 entry count, then the raw u32 entries.
// The table goes to `path` through `path`.partial + rename: a file of that name is always complete.
// Chunks are staged through two page-locked buffers so that the download of one overlaps the write
// of the other.  err receives the message (the caller's thread sets it as its last error).
static int k15_write_file_on(int device, hipStream_t stream, const uint32_t *d_table, const char *path, std::string &err)
{
    const std::string tmp = std::string(path) + ".partial";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) {
        err = std::string("cannot open ") + tmp + " for writing";
        return LRB_ERR_IO;
    }
    const uint64_t entries = LRB_K15_ENTRIES;
    int rc = LRB_OK;
    const uint64_t chunk = 32ull << 20; // entries per staged chunk (128 MiB)
    uint32_t *h[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    if (hipSetDevice(device) != hipSuccess || hipHostMalloc((void **)&h[0], chunk * 4, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void **)&h[1], chunk * 4, hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&ev[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ev[1], hipEventDisableTiming) != hipSuccess) {
        rc = LRB_ERR_NOMEM;
        err = "pinned staging allocation failed";
    }
    if (rc == LRB_OK && fwrite(&entries, 8, 1, f) != 1) rc = LRB_ERR_IO;
    const uint64_t n_chunks = entries / chunk;
    auto fetch = [&](uint64_t i) {
        return hipMemcpyAsync(h[i & 1], d_table + i * chunk, chunk * 4, hipMemcpyDeviceToHost, stream) == hipSuccess &&
               hipEventRecord(ev[i & 1], stream) == hipSuccess;
    };
    if (rc == LRB_OK && !fetch(0)) rc = LRB_ERR_HIP;
    for (uint64_t i = 0; rc == LRB_OK && i < n_chunks; ++i) {
        if (i + 1 < n_chunks && !fetch(i + 1)) rc = LRB_ERR_HIP;
        if (rc == LRB_OK && hipEventSynchronize(ev[i & 1]) != hipSuccess) rc = LRB_ERR_HIP;
        if (rc == LRB_OK && fwrite(h[i & 1], 4, chunk, f) != chunk) rc = LRB_ERR_IO;
    }
    if (rc == LRB_ERR_HIP) {
        err = "table download failed";
        (void)hipStreamSynchronize(stream);
    }
    for (int i = 0; i < 2; ++i) {
        if (h[i]) (void)hipHostFree(h[i]);
        if (ev[i]) (void)hipEventDestroy(ev[i]);
    }
    if (fclose(f) != 0) rc = rc == LRB_OK ? LRB_ERR_IO : rc;
    if (rc == LRB_OK && rename(tmp.c_str(), path) != 0) rc = LRB_ERR_IO;
    if (rc == LRB_ERR_IO && err.empty()) err = std::string("write to ") + path + " failed";
    if (rc != LRB_OK) remove(tmp.c_str());
    return rc;
}

extern "C" int lrb_k15_write_file(lrb_ctx *c, const uint32_t *d_table, const char *path)
{
    ARG_TRY(c != nullptr && d_table != nullptr && path != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    std::string err;
    const int rc = k15_write_file_on(c->device, c->stream, d_table, path, err);
    if (rc != LRB_OK) lrb_set_error("%s%s", err.c_str(), "");
    return rc;
}

// The same on a thread and a stream of its own: the caller goes on (coverage, VAE) while 4 GiB go to
// the file.  The table must stay allocated and unchanged until lrb_job_wait.
struct lrb_job {
    std::thread th;
    hipStream_t stream = nullptr;
    int rc = LRB_OK;
    std::string err;
};

extern "C" int lrb_k15_write_file_async(lrb_ctx *c, const uint32_t *d_table, const char *path, lrb_job **out)
{
    ARG_TRY(c != nullptr && d_table != nullptr && path != nullptr && out != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream)); // the table is final
    lrb_job *job = new (std::nothrow) lrb_job();
    if (!job) return LRB_ERR_NOMEM;
    if (hipStreamCreateWithFlags(&job->stream, hipStreamNonBlocking) != hipSuccess) {
        delete job;
        lrb_set_error("cannot create a stream for the table writer%s%s", "", "");
        return LRB_ERR_HIP;
    }
    const int device = c->device;
    const std::string p(path);
    try {
        job->th = std::thread([job, device, d_table, p]() { job->rc = k15_write_file_on(device, job->stream, d_table, p.c_str(), job->err); });
    } catch (...) {
        (void)hipStreamDestroy(job->stream);
        delete job;
        lrb_set_error("cannot start the table writer thread%s%s", "", "");
        return LRB_ERR_NOMEM;
    }
    *out = job;
    return LRB_OK;
}

// ONE part of the table file, for writers that share the work (the ranks of the sharded driver all hold the whole
// table after the all-reduce): entries [part E / n_parts, (part + 1) E / n_parts) go to their place -- byte 8 + 4 first --
// of the EXISTING file `path` (made at its full size by one of the writers, e.g. ftruncate; no rename here: the callers
// agree on when the file is complete); part 0 also writes the entry count in front.  Same staging as the whole-file writer.
static int k15_write_part_on(int device, hipStream_t stream, const uint32_t *d_table, const char *path, uint32_t part, uint32_t n_parts,
                             std::string &err)
{
    const uint64_t entries = LRB_K15_ENTRIES, first = entries * part / n_parts, last = entries * (part + 1) / n_parts;
    const int fd = open(path, O_WRONLY);
    if (fd < 0) {
        err = std::string("cannot open ") + path + " for writing";
        return LRB_ERR_IO;
    }
    int rc = LRB_OK;
    const uint64_t chunk = 32ull << 20; // entries per staged chunk (128 MiB)
    uint32_t *h[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    if (hipSetDevice(device) != hipSuccess || hipHostMalloc((void **)&h[0], chunk * 4, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void **)&h[1], chunk * 4, hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&ev[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ev[1], hipEventDisableTiming) != hipSuccess) {
        rc = LRB_ERR_NOMEM;
        err = "pinned staging allocation failed";
    }
    auto put = [&](const void *buf, uint64_t bytes, uint64_t at) {
        const char *b = (const char *)buf;
        while (bytes) {
            const ssize_t w = pwrite(fd, b, bytes, (off_t)at);
            if (w <= 0) return false;
            b += w;
            at += (uint64_t)w;
            bytes -= (uint64_t)w;
        }
        return true;
    };
    if (rc == LRB_OK && part == 0 && !put(&entries, 8, 0)) rc = LRB_ERR_IO;
    const uint64_t n_chunks = (last - first + chunk - 1) / chunk;
    auto len_of = [&](uint64_t i) { return first + (i + 1) * chunk <= last ? chunk : last - first - i * chunk; };
    auto fetch = [&](uint64_t i) {
        return hipMemcpyAsync(h[i & 1], d_table + first + i * chunk, len_of(i) * 4, hipMemcpyDeviceToHost, stream) == hipSuccess &&
               hipEventRecord(ev[i & 1], stream) == hipSuccess;
    };
    if (rc == LRB_OK && n_chunks && !fetch(0)) rc = LRB_ERR_HIP;
    for (uint64_t i = 0; rc == LRB_OK && i < n_chunks; ++i) {
        if (i + 1 < n_chunks && !fetch(i + 1)) rc = LRB_ERR_HIP;
        if (rc == LRB_OK && hipEventSynchronize(ev[i & 1]) != hipSuccess) rc = LRB_ERR_HIP;
        if (rc == LRB_OK && !put(h[i & 1], len_of(i) * 4, 8 + 4 * (first + i * chunk))) rc = LRB_ERR_IO;
    }
    if (rc == LRB_ERR_HIP) {
        err = "table download failed";
        (void)hipStreamSynchronize(stream);
    }
    for (int i = 0; i < 2; ++i) {
        if (h[i]) (void)hipHostFree(h[i]);
        if (ev[i]) (void)hipEventDestroy(ev[i]);
    }
    if (close(fd) != 0) rc = rc == LRB_OK ? LRB_ERR_IO : rc;
    if (rc == LRB_ERR_IO && err.empty()) err = std::string("write to ") + path + " failed";
    return rc;
}

extern "C" int lrb_k15_write_file_part_async(lrb_ctx *c, const uint32_t *d_table, const char *path, uint32_t part, uint32_t n_parts,
                                             lrb_job **out)
{
    ARG_TRY(c != nullptr && d_table != nullptr && path != nullptr && out != nullptr && n_parts >= 1 && part < n_parts);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream)); // the table is final
    lrb_job *job = new (std::nothrow) lrb_job();
    if (!job) return LRB_ERR_NOMEM;
    if (hipStreamCreateWithFlags(&job->stream, hipStreamNonBlocking) != hipSuccess) {
        delete job;
        lrb_set_error("cannot create a stream for the table writer%s%s", "", "");
        return LRB_ERR_HIP;
    }
    const int device = c->device;
    const std::string p(path);
    try {
        job->th = std::thread(
            [job, device, d_table, p, part, n_parts]() { job->rc = k15_write_part_on(device, job->stream, d_table, p.c_str(), part, n_parts, job->err); });
    } catch (...) {
        (void)hipStreamDestroy(job->stream);
        delete job;
        lrb_set_error("cannot start the table writer thread%s%s", "", "");
        return LRB_ERR_NOMEM;
    }
    *out = job;
    return LRB_OK;
}

extern "C" int lrb_job_wait(lrb_job *job)
{
    if (!job) return LRB_OK;
    if (job->th.joinable()) job->th.join();
    const int rc = job->rc;
    if (rc != LRB_OK) lrb_set_error("%s%s", job->err.c_str(), "");
    if (job->stream) (void)hipStreamDestroy(job->stream);
    delete job;
    return rc;
}

extern "C" int lrb_k15_read_file(lrb_ctx *c, uint32_t *d_table, const char *path)
{
    ARG_TRY(c != nullptr && d_table != nullptr && path != nullptr);
    HIP_TRY(hipSetDevice(c->device));
    FILE *f = fopen(path, "rb");
    if (!f) {
        lrb_set_error("cannot open %s%s", path, "");
        return LRB_ERR_IO;
    }
    uint64_t entries = 0;
    if (fread(&entries, 8, 1, f) != 1 || entries != LRB_K15_ENTRIES) {
        fclose(f);
        lrb_set_error("%s: not a 15-mer table (bad entry count)%s", path, "");
        return LRB_ERR_FORMAT;
    }
    const uint64_t chunk = 64ull << 20;
    uint32_t *h = nullptr;
    if (hipHostMalloc((void **)&h, chunk * 4, hipHostMallocDefault) != hipSuccess) {
        fclose(f);
        lrb_set_error("pinned staging allocation failed%s%s", "", "");
        return LRB_ERR_NOMEM;
    }
    int rc = LRB_OK;
    for (uint64_t s = 0; rc == LRB_OK && s < entries; s += chunk) {
        if (fread(h, 4, chunk, f) != chunk) {
            rc = LRB_ERR_FORMAT;
            lrb_set_error("%s: truncated table%s", path, "");
            break;
        }
        if (hipMemcpyAsync(d_table + s, h, chunk * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess) {
            rc = LRB_ERR_HIP;
            lrb_set_error("table upload failed%s%s", "", "");
        }
    }
    (void)hipHostFree(h);
    fclose(f);
    return rc;
}
