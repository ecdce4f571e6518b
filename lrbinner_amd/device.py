"""Python face of the C ABI: a ``Context`` per GPU plus thin typed wrappers.

Two levels, as in include/lrb_hip.h:

* host level  -- numpy in, numpy out (``kmer_counts``, ``k15_accumulate``,
  ``cov_hist``): the library uploads, packs, runs the kernels and downloads.
* device level -- torch CUDA tensors in, torch CUDA tensors out (``pack``,
  ``kmer_counts_dev`` ...): nothing leaves HBM.  torch is only the allocator
  and the stream here; the arithmetic is in liblrb_hip.so.
"""
import ctypes as C
import os
import sys
import time

import numpy as np

from . import _lib
from ._lib import K15_ENTRIES, K15_HALF_ENTRIES, HIST_BINS, call, lib, u8p, u32p, u64p, f64p, vp


def _np(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


def _ptr(a, t):
    return a.ctypes.data_as(t)


def kmer_dim(k):
    d = C.c_uint32(0)
    call("lrb_kmer_dim", int(k), C.byref(d))
    return d.value


def kmer_lut(k):
    """Canonical index of every k-mer code (count-kmers.cpp:38-64)."""
    lut = np.zeros(4 ** k, dtype=np.uint32)
    d = C.c_uint32(0)
    call("lrb_kmer_lut", int(k), _ptr(lut, u32p), C.byref(d))
    return lut, d.value


def pack_layout(offs):
    """offsets[n+1] -> (lens u32[n], code_off u64[n+1], mask_off u64[n+1])."""
    offs = _np(offs, np.uint64)
    n = len(offs) - 1
    lens = np.zeros(max(n, 1), dtype=np.uint32)
    co = np.zeros(n + 1, dtype=np.uint64)
    mo = np.zeros(n + 1, dtype=np.uint64)
    call("lrb_pack_layout", _ptr(offs, u64p), n, _ptr(lens, u32p), _ptr(co, u64p), _ptr(mo, u64p))
    return lens[:n], co, mo


class PackedReads:
    """Reads resident in HBM in the packed layout (torch tensors own the memory)."""

    def __init__(self, codes, mask, code_off, mask_off, lens, n, planes=None):
        self.codes, self.mask = codes, mask
        self.code_off, self.mask_off, self.lens = code_off, mask_off, lens
        self.n = n
        self.planes = planes  # bit-plane form for the k=3 kernel (optional)
        self.planes_t = None  # group-transposed bit planes (lane-per-read kernel)
        self.group_off = None
        self.order = None     # slot -> read of the transposed layout (None: identity)
        self.codes_t = None   # group-transposed 2-bit codes (lane-per-read k=4 kernel)
        self.group_off4 = None
        self.order4 = None


class WindowLists:
    """The windows of a set of resident reads partitioned by map bucket per group of reads (Context.lists_part_dev):
    what K2's tally and K3's sweep both start from.  torch tensors own the memory: the lists, bounds[g][16385] (where
    each bucket starts in its group's region) and gbase[g] (where the regions start)."""
    pr = lists = bounds = gbase = None
    R = ngroups = bins = n = words = 0


class ResidentBatch:
    """A batch of reads packed in HBM (lrb_packed): the three profile stages run on it
    without re-reading the file or re-crossing PCIe."""

    def __init__(self, ctx, handle, lens):
        self.ctx, self._h, self.lens = ctx, handle, lens
        n, b = C.c_uint64(0), C.c_uint64(0)
        call("lrb_packed_info", self._h, C.byref(n), C.byref(b))
        self.n, self.device_bytes = n.value, b.value
        self.total_bases = int(np.asarray(lens, dtype=np.uint64).sum())

    def kmer_counts(self, k):
        out = np.zeros((self.n, kmer_dim(k)), dtype=np.uint32)
        call("lrb_packed_kmer_counts", self.ctx._h, self._h, int(k), _ptr(out, u32p))
        return out

    def kmer_counts_dev(self, k, out_ptr):
        """The kernel half of kmer_counts / kmer_text: uint32[n, dim] tallies into device memory at out_ptr."""
        call("lrb_packed_kmer_counts_dev", self.ctx._h, self._h, int(k), vp(out_ptr))

    def k15_accumulate(self, table_ptr):
        call("lrb_packed_k15_accumulate", self.ctx._h, self._h, vp(table_ptr))

    def k15_accumulate_half(self, half_ptr):
        """One atomic per window into the canonical half of the table (small batches)."""
        call("lrb_packed_k15_accumulate_half", self.ctx._h, self._h, vp(half_ptr))

    def cov_hist(self, table_ptr, bin_size, bins):
        hist = np.zeros((self.n, max(int(bins), 0)), dtype=np.uint32)
        sums = np.zeros(self.n, dtype=np.uint32)
        call("lrb_packed_cov_hist", self.ctx._h, self._h, vp(table_ptr), int(bin_size), int(bins),
             _ptr(hist, u32p), _ptr(sums, u32p))
        return hist, sums

    def kmer_text(self, k, want_q=True, slot=0):
        """com_profs rows of the batch (uint8 array, fixed-width rows, formatted on the device)
        [+ the six-decimal integers: the text parses to q / 1e6].  The arrays live in the
        context's page-locked staging and are overwritten by the next *_text call with the
        same ``slot`` (two slots let a writer thread drain one while the next batch fills the other)."""
        dim = kmer_dim(k)
        # page-locked and reused: valid until the next *_text call on this context
        text = self.ctx.pinned(f"text{slot}", self.n * int(lib().lrb_com_row_bytes(dim)))
        q = self.ctx.pinned(f"q6{slot}", 4 * self.n * dim, np.uint32).reshape(self.n, dim) if want_q else None
        call("lrb_packed_kmer_text", self.ctx._h, self._h, int(k), vp(text.ctypes.data),
             _ptr(q, u32p) if want_q else None)
        return (text, q) if want_q else text

    def cov_text(self, table_ptr, bin_size, bins, want_q=True, slot=0):
        """cov_profs rows of the batch, as kmer_text."""
        bins = int(bins)
        text = self.ctx.pinned(f"text{slot}", self.n * int(lib().lrb_cov_row_bytes(max(bins, 0))))
        q = self.ctx.pinned(f"q6{slot}", 4 * self.n * max(bins, 0), np.uint32).reshape(self.n, max(bins, 0)) if want_q else None
        call("lrb_packed_cov_text", self.ctx._h, self._h, vp(table_ptr), int(bin_size), bins,
             vp(text.ctypes.data), _ptr(q, u32p) if want_q else None)
        return (text, q) if want_q else text

    def free(self):
        if self._h:
            lib().lrb_packed_free(self.ctx._h, self._h)
            self._h = vp()


class PackedLists:
    """The windows of several resident batches partitioned once (lrb_winlists): K2's tally into the canonical half of
    the table and K3's sweep both start from it.  Owns a copy of the packed reads and the slice lists (4.4 bytes per
    base); keep it between the two stages while memory allows."""

    def __init__(self, ctx, batches, bins=32, workspace=False):
        """workspace=True: the buffers are the context's workspaces -- no allocation (16 GB of hipMalloc is 0.4 s), valid
        until the next call that uses those workspaces (valid())."""
        self.ctx, self.batches = ctx, list(batches)
        arr = (vp * max(len(self.batches), 1))(*[b._h for b in self.batches])
        self._h = vp()
        call("lrb_packed_lists_create", ctx._h, arr, len(self.batches), int(bins), 1 if workspace else 0, C.byref(self._h))
        n, b, r = C.c_uint64(0), C.c_uint64(0), C.c_uint32(0)
        call("lrb_winlists_info", self._h, C.byref(n), C.byref(b), C.byref(r))
        self.n, self.device_bytes, self.reads_per_group = n.value, b.value, r.value

    def valid(self):
        v = C.c_int(0)
        call("lrb_winlists_valid", self.ctx._h, self._h, C.byref(v))
        return bool(v.value)

    def fits(self, bins):
        """Can these lists be swept for a histogram of `bins` bins (group's u16 counters within 127 KB of LDS)?"""
        # (two reads' u16 counters share a word: an even number of reads must fit -- lrb_wl_hist_fits in lrb_lists.hip)
        return 1 <= int(bins) <= 256 and ((self.reads_per_group + 1) & ~1) * int(bins) <= 65024

    def tally(self, half_ptr):
        call("lrb_winlists_tally", self.ctx._h, self._h, vp(half_ptr))

    def cov_hist(self, map_ptr, bins):
        """K3 as a sweep of these lists (the kernel half of cov_text): the histograms stay in the context for cov_rows."""
        call("lrb_winlists_cov_hist", self.ctx._h, self._h, vp(map_ptr), int(bins))

    def cov_text(self, map_ptr, bins, want_q=True, slot=0, chunk_rows=None):
        """K3 as a sweep of these lists, then the cov_profs rows of every batch in turn (as Context.cov_text_many)."""
        self.cov_hist(map_ptr, bins)
        yield from self.ctx.cov_rows(self.batches, int(bins), want_q, slot, chunk_rows)

    def free(self):
        if self._h:
            lib().lrb_winlists_free(self.ctx._h, self._h)
            self._h = vp()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """One GPU, one HIP stream.  ``stream=None`` -> the library makes its own;
    ``use_torch_stream=True`` enqueues on torch's current stream so torch ops and
    library kernels are ordered without extra synchronisation."""

    def __init__(self, device=0, use_torch_stream=False):
        self._h = vp()
        stream = None
        if use_torch_stream:
            import torch
            stream = vp(torch.cuda.current_stream(device).cuda_stream)
        call("lrb_ctx_create", int(device), stream, 0 if use_torch_stream else 1,
             C.byref(self._h))
        self.device = int(device)

    def close(self):
        if self._h:
            for ptr, _ in getattr(self, "_pin", {}).values():
                lib().lrb_host_free(self._h, vp(ptr))
            self._pin = {}
            lib().lrb_ctx_destroy(self._h)
            self._h = vp()

    def pinned(self, key, nbytes, dtype=np.uint8):
        """A page-locked host array of ``nbytes`` bytes, reused (and overwritten) by the next
        request with the same key: D2H copies into it run at link speed and a fresh
        allocation's page faults are paid once."""
        pin = self.__dict__.setdefault("_pin", {})
        ent = pin.get(key)
        if ent is None or ent[1] < nbytes:
            if ent is not None:
                call("lrb_host_free", self._h, vp(ent[0]))
                del pin[key]
            cap = max(int(nbytes) * 5 // 4, 1 << 20)
            p = vp()
            call("lrb_host_alloc", self._h, cap, C.byref(p))
            ent = pin[key] = (p.value, cap)
        raw = (C.c_uint8 * int(nbytes)).from_address(ent[0])
        return np.frombuffer(raw, dtype=np.uint8).view(dtype)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def sync(self):
        call("lrb_ctx_sync", self._h)

    def trim(self, keep_below=1 << 28):
        """Free the context's workspaces of keep_below bytes and more (K2 partition buffers)."""
        call("lrb_ctx_trim", self._h, int(keep_below))

    def list_pool(self, max_bytes):
        """Retain up to max_bytes of the memory freed slice lists give back for the next ones (lrb_ctx_list_pool):
        for hosts that keep lists call after call; 0 turns it off and frees what is retained."""
        call("lrb_ctx_list_pool", self._h, int(max_bytes))

    # ---------------- raw device memory (no torch needed) -----------------
    def alloc(self, nbytes):
        p = vp()
        call("lrb_dev_alloc", self._h, int(nbytes), C.byref(p))
        return p.value

    def mem_info(self):
        """(free, total) bytes of device memory."""
        f, t = C.c_uint64(0), C.c_uint64(0)
        call("lrb_dev_mem_info", self._h, C.byref(f), C.byref(t))
        return f.value, t.value

    def free(self, ptr):
        call("lrb_dev_free", self._h, vp(ptr))

    def memset(self, ptr, value, nbytes):
        call("lrb_dev_memset", self._h, vp(ptr), int(value), int(nbytes))

    def h2d(self, ptr, arr):
        arr = np.ascontiguousarray(arr)
        call("lrb_copy_h2d", self._h, vp(ptr), vp(arr.ctypes.data), arr.nbytes)

    def d2h(self, arr, ptr):
        assert arr.flags["C_CONTIGUOUS"]
        call("lrb_copy_d2h", self._h, vp(arr.ctypes.data), vp(ptr), arr.nbytes)

    def k15_accumulate_many(self, batches, table_ptr):
        """FORWARD tallies of many ResidentBatch objects into a full table (lrb_packed_k15_accumulate_many: one atomic
        a window; the product's K2 is k15_tally_half_many)."""
        batches = list(batches)
        arr = (vp * max(len(batches), 1))(*[b._h for b in batches])
        call("lrb_packed_k15_accumulate_many", self._h, arr, len(batches), vp(table_ptr))

    def partition_retries(self):
        """Partitions of window lists this context has repeated (lrb_ctx_partition_retries): 0 on a GPU of its own."""
        n = C.c_uint64(0)
        call("lrb_ctx_partition_retries", self._h, C.byref(n))
        return n.value

    def kmer_counts_many_dev(self, batches, k, out_ptr):
        """K1 of several ResidentBatch objects, rows in batch order at the device address out_ptr (sum of n x dim uint32):
        one launch over all their groups for k = 4 (lrb_packed_kmer_counts_many_dev)."""
        batches = list(batches)
        arr = (vp * max(len(batches), 1))(*[b._h for b in batches])
        call("lrb_packed_kmer_counts_many_dev", self._h, arr, len(batches), int(k), vp(out_ptr))

    def k15_tally_half_many(self, batches, half_ptr, bins=32):
        """K2 of many ResidentBatch objects into the canonical half of the table, groups of batches sharing one
        partition of their windows in the context's workspaces (lrb_packed_k15_tally_half_many_for).  ``bins``: the
        histogram width of the coverage stage to follow -- the LAST group's lists stay in the workspaces for it
        (lists_resident / cov_hist_many of that group)."""
        batches = list(batches)
        arr = (vp * max(len(batches), 1))(*[b._h for b in batches])
        bins = int(bins) if bins and 1 <= int(bins) <= 256 else 32
        call("lrb_packed_k15_tally_half_many_for", self._h, arr, len(batches), vp(half_ptr), bins)

    def lists_resident(self, batches, bins):
        """Do the workspaces still hold the slice lists of exactly these batches (the last group of
        k15_tally_half_many), fit for a histogram of ``bins`` bins?  cov_hist_many of them is then the sweep alone."""
        batches = list(batches)
        if not batches or not 1 <= int(bins) <= 256 or not all(b._h for b in batches):
            return False
        arr = (vp * len(batches))(*[b._h for b in batches])
        yes = C.c_int(0)
        call("lrb_packed_lists_resident", self._h, arr, len(batches), int(bins), C.byref(yes))
        return bool(yes.value)

    def cov_map_build(self, table_ptr, bin_size, bins):
        """Compact map of a finished table (raw device pointers): 2^29 bytes, one bin id per pair (x, rc(x)).
        Returns the map's device pointer; the caller frees it."""
        m = self.alloc(K15_HALF_ENTRIES)
        try:
            call("lrb_cov_map_build_dev", self._h, vp(table_ptr), int(bin_size), int(bins), vp(m))
        except BaseException:
            self.free(m)
            raise
        return m

    def cov_map_build_half(self, half_ptr, bin_size, bins):
        """The same map from the canonical half of the table (raw device pointers); the caller frees it."""
        m = self.alloc(K15_HALF_ENTRIES)
        try:
            call("lrb_cov_map_build_half_dev", self._h, vp(half_ptr), int(bin_size), int(bins), vp(m))
        except BaseException:
            self.free(m)
            raise
        return m

    def cov_hist_many(self, batches, map_ptr, bins):
        """K3 of several resident batches as ONE sweep against the compact map (lrb_packed_cov_hist_many: windows
        partitioned, then swept): the kernel half of cov_text_many; the histograms stay in the context for cov_rows."""
        bins = int(bins)
        arr = (vp * len(batches))(*[b._h for b in batches])
        t0 = time.perf_counter()
        call("lrb_packed_cov_hist_many", self._h, arr, len(batches), vp(map_ptr), bins)
        if os.environ.get("LRB_TIMING"):
            self.sync()
            print(f"[timing] cov_hist_many: {len(batches)} batches, {sum(b.n for b in batches)} reads, "
                  f"{(time.perf_counter() - t0) * 1e3:.1f} ms", file=sys.stderr, flush=True)

    def cov_text_many(self, batches, map_ptr, bins, want_q=True, slot=0, chunk_rows=None):
        """cov_hist_many, then the cov_profs rows of every batch in turn: yields (slot, text, q6) per batch, in the
        page-locked staging of ``slot()`` -- a callable giving the slot to format the next batch into (see
        ResidentBatch.kmer_text)."""
        self.cov_hist_many(batches, map_ptr, bins)
        yield from self.cov_rows(batches, int(bins), want_q, slot, chunk_rows)

    def cov_rows(self, batches, bins, want_q=True, slot=0, chunk_rows=None):
        """cov_profs rows of the histograms a many-batch K3 call left in the context, batch by batch -- or, with
        chunk_rows, in pieces of that many rows whatever the batches (a caller that appends to one file: 746 calls of
        2.4 MB each were 0.4 s of latencies at C3's size, 80 calls of 19 MB are not)."""
        row = 0
        width = int(lib().lrb_cov_row_bytes(bins))
        if chunk_rows:
            total = sum(b.n for b in batches)
            while row < total:
                n = min(int(chunk_rows), total - row)
                s_ = slot() if callable(slot) else slot
                text = self.pinned(f"text{s_}", n * width)
                q = self.pinned(f"q6{s_}", 4 * n * bins, np.uint32).reshape(n, bins) if want_q else None
                call("lrb_cov_rows_text", self._h, row, n, bins, vp(text.ctypes.data), _ptr(q, u32p) if want_q else None)
                row += n
                yield s_, text, q
            return
        for b in batches:
            s_ = slot() if callable(slot) else slot
            text = self.pinned(f"text{s_}", b.n * width)
            q = self.pinned(f"q6{s_}", 4 * b.n * bins, np.uint32).reshape(b.n, bins) if want_q else None
            call("lrb_cov_rows_text", self._h, row, b.n, bins, vp(text.ctypes.data), _ptr(q, u32p) if want_q else None)
            row += b.n
            yield s_, text, q

    def alloc_half(self):
        """A zeroed canonical half of the 15-mer table (2^29 uint32, 2 GiB); returns the device pointer."""
        p = self.alloc(4 * K15_HALF_ENTRIES)
        self.memset(p, 0, 4 * K15_HALF_ENTRIES)
        self.sync()
        return p

    def k15_expand_half(self, half_ptr, table_ptr):
        """table[x] = table[rc(x)] = half[h(x)] (raw device pointers)."""
        call("lrb_k15_expand_half_dev", self._h, vp(half_ptr), vp(table_ptr))
        self.sync()

    def alloc_table(self):
        """A zeroed 4^15-entry uint32 table (4 GiB); returns the device pointer."""
        p = self.alloc(4 * K15_ENTRIES)
        self.memset(p, 0, 4 * K15_ENTRIES)
        self.sync()
        return p

    # ---------------- host level ------------------------------------------
    def kmer_counts(self, seqs, offs, k):
        """uint32[n, dim] canonical k-mer tallies (integer view of count_kmers)."""
        seqs, offs = _np(seqs, np.uint8), _np(offs, np.uint64)
        n = len(offs) - 1
        out = np.zeros((n, kmer_dim(k)), dtype=np.uint32)
        call("lrb_kmer_counts_host", self._h, _ptr(seqs, u8p), _ptr(offs, u64p), n, int(k),
             _ptr(out, u32p))
        return out

    def k15_accumulate(self, seqs, offs, table_ptr):
        seqs, offs = _np(seqs, np.uint8), _np(offs, np.uint64)
        call("lrb_k15_accumulate_host", self._h, _ptr(seqs, u8p), _ptr(offs, u64p),
             len(offs) - 1, vp(table_ptr))

    def k15_mirror(self, table_ptr):
        call("lrb_k15_mirror_dev", self._h, vp(table_ptr))
        self.sync()

    def k15_write_file(self, table_ptr, path):
        call("lrb_k15_write_file", self._h, vp(table_ptr), os.fsencode(path))

    def k15_write_file_async(self, table_ptr, path):
        """Start writing the table file on the library's own thread and stream; returns a job for
        job_wait.  The table must stay allocated and unchanged until then."""
        job = vp()
        call("lrb_k15_write_file_async", self._h, vp(table_ptr), os.fsencode(path), C.byref(job))
        return job

    def k15_write_file_part_async(self, table_ptr, path, part, n_parts):
        """Start writing part `part` of `n_parts` of the table into the EXISTING file `path` (full size already) on the
        library's own thread and stream; returns a job for job_wait (lrb_k15_write_file_part_async)."""
        job = vp()
        call("lrb_k15_write_file_part_async", self._h, vp(table_ptr), os.fsencode(path), int(part), int(n_parts), C.byref(job))
        return job

    @staticmethod
    def job_wait(job):
        call("lrb_job_wait", job)

    def k15_read_file(self, table_ptr, path):
        call("lrb_k15_read_file", self._h, vp(table_ptr), os.fsencode(path))

    def cov_hist(self, seqs, offs, table_ptr, bin_size, bins):
        """(hist uint32[n, bins], sums uint32[n]) -- integer view of line_to_vec."""
        seqs, offs = _np(seqs, np.uint8), _np(offs, np.uint64)
        n = len(offs) - 1
        hist = np.zeros((n, max(int(bins), 0)), dtype=np.uint32)
        sums = np.zeros(n, dtype=np.uint32)
        call("lrb_cov_hist_host", self._h, _ptr(seqs, u8p), _ptr(offs, u64p), n, vp(table_ptr),
             int(bin_size), int(bins), _ptr(hist, u32p), _ptr(sums, u32p))
        return hist, sums

    # ---------------- resident batches (no torch needed) --------------------
    def packed_create(self, seqs, offs, with_planes=True):
        """Upload + pack one batch and keep it in HBM.  Returns a ResidentBatch.
        with_planes: False / True (the k = 3 layout) or the bit set of include/lrb_hip.h
        (1 group-transposed bit planes for k = 3, 2 group-transposed codes for k = 4, 5)."""
        seqs, offs = _np(seqs, np.uint8), _np(offs, np.uint64)
        h = vp()
        call("lrb_packed_create", self._h, _ptr(seqs, u8p), _ptr(offs, u64p), len(offs) - 1,
             int(with_planes) & 3, C.byref(h))
        return ResidentBatch(self, h, np.diff(offs).astype(np.uint32))

    def packed_create_packed(self, hp, with_planes=True):
        """As packed_create for a batch the HOST has packed already (HostPacked: pack_reads_host, or
        ParallelReader.next_packed): codes, masks and offsets are uploaded as they are -- 0.375 bytes a base over PCIe --
        and the transposed layouts are made from the codes on the device (lrb_packed_create_packed)."""
        h = vp()
        call("lrb_packed_create_packed", self._h, _ptr(hp.codes, u32p), _ptr(hp.mask, u32p), _ptr(hp.code_off, u64p),
             _ptr(hp.mask_off, u64p), _ptr(hp.lens, u32p), _ptr(hp.offs, u64p), hp.n, int(with_planes) & 3, C.byref(h))
        return ResidentBatch(self, h, np.array(hp.lens[:hp.n], dtype=np.uint32))

    def packed_create_dev(self, seqs_ptr, offs, with_planes=True):
        """As packed_create for bases ALREADY in HBM (seqs_ptr: device address of the byte that offs indexes from;
        offs a host array): only the offsets and lengths cross PCIe (lrb_packed_create_dev)."""
        offs = _np(offs, np.uint64)
        h = vp()
        call("lrb_packed_create_dev", self._h, vp(seqs_ptr), _ptr(offs, u64p), len(offs) - 1, int(with_planes) & 3, C.byref(h))
        return ResidentBatch(self, h, np.diff(offs).astype(np.uint32))

    # ---------------- device level (torch tensors) --------------------------
    def pack(self, seqs_t, offs, want_mask=True, want_planes=False):
        """ASCII bytes already in HBM (uint8 CUDA tensor) -> PackedReads."""
        import torch
        offs = _np(offs, np.uint64)
        n = len(offs) - 1
        lens, co, mo = pack_layout(offs)
        dev = seqs_t.device
        t = lambda a, dt: torch.from_numpy(a.view(dt)).to(dev)
        offs_t, co_t, mo_t = t(offs, np.int64), t(co, np.int64), t(mo, np.int64)
        lens_t = t(lens if n else np.zeros(1, np.uint32), np.int32)
        codes = torch.empty(int(co[-1]) or 4, dtype=torch.int32, device=dev)
        mask = torch.empty(int(mo[-1]) or 4, dtype=torch.int32, device=dev) if want_mask else None
        planes = (torch.empty(2 * int(mo[-1]) or 8, dtype=torch.int32, device=dev)
                  if want_planes else None)
        call("lrb_pack_reads_dev", self._h, vp(seqs_t.data_ptr()), int(offs[-1]),
             vp(offs_t.data_ptr()), n, vp(co_t.data_ptr()), vp(mo_t.data_ptr()),
             vp(codes.data_ptr()), vp(mask.data_ptr()) if want_mask else None,
             vp(planes.data_ptr()) if want_planes else None)
        self.sync()
        return PackedReads(codes, mask, co_t, mo_t, lens_t, n, planes)

    def kmer_counts_dev(self, pr, k, out=None):
        import torch
        dim = kmer_dim(k)
        if out is None:
            out = torch.empty((pr.n, dim), dtype=torch.int32, device=pr.codes.device)
        call("lrb_kmer_counts_dev", self._h, vp(pr.codes.data_ptr()), vp(pr.code_off.data_ptr()),
             vp(pr.lens.data_ptr()), pr.n, int(k), vp(out.data_ptr()))
        return out

    def make_planes(self, pr):
        """Derive the bit-plane form of the reads (k=3 kernel) from the packed codes."""
        import torch
        nwords = 2 * int(pr.mask_off[-1].item())
        pr.planes = torch.empty(max(nwords, 8), dtype=torch.int32, device=pr.codes.device)
        call("lrb_planes_from_codes_dev", self._h, vp(pr.codes.data_ptr()),
             vp(pr.code_off.data_ptr()), vp(pr.mask_off.data_ptr()), pr.n,
             vp(pr.planes.data_ptr()))
        return pr.planes

    def _planes_t_layout(self, pr, sort):
        import torch
        lens = pr.lens.cpu().numpy().view(np.uint32)[: pr.n].copy()
        goff = np.zeros((pr.n + 63) // 64 + 1, dtype=np.uint64)
        order = np.zeros(max(pr.n, 1), dtype=np.uint32) if sort else None
        call("lrb_planes_t_layout", _ptr(lens, u32p), pr.n, _ptr(order, u32p) if sort else None,
             _ptr(goff, u64p))
        dev = pr.lens.device
        pr.group_off = torch.from_numpy(goff.view(np.int64)).to(dev)
        pr.order = torch.from_numpy(order.view(np.int32)).to(dev) if sort else None
        pr.planes_t = torch.empty(max(int(goff[-1]) * 128, 8), dtype=torch.int32, device=dev)
        return vp(pr.order.data_ptr()) if sort else None

    def make_planes_t(self, pr, sort=True):
        """Group-transposed bit planes for the lane-per-read k=3 kernel, from pr.planes."""
        if pr.planes is None:
            self.make_planes(pr)
        o = self._planes_t_layout(pr, sort)
        call("lrb_planes_t_from_planes_dev", self._h, vp(pr.planes.data_ptr()),
             vp(pr.mask_off.data_ptr()), vp(pr.group_off.data_ptr()), o, pr.n,
             vp(pr.planes_t.data_ptr()))
        return pr.planes_t

    def pack_planes_t(self, seqs_t, offs, sort=True):
        """ASCII in HBM -> PackedReads holding only the group-transposed planes (k=3)."""
        import torch
        offs = _np(offs, np.uint64)
        n = len(offs) - 1
        dev = seqs_t.device
        lens = np.diff(offs).astype(np.uint32) if n else np.zeros(1, np.uint32)
        pr = PackedReads(None, None, None, None,
                         torch.from_numpy(lens.view(np.int32)).to(dev), n)
        o = self._planes_t_layout(pr, sort)
        offs_t = torch.from_numpy(offs.view(np.int64)).to(dev)
        call("lrb_pack_planes_t_dev", self._h, vp(seqs_t.data_ptr()), vp(offs_t.data_ptr()),
             vp(pr.group_off.data_ptr()), o, n, vp(pr.planes_t.data_ptr()))
        self.sync()
        return pr

    def kmer_counts3t_dev(self, pr, out=None):
        """k=3 tallies by the lane-per-read kernel on pr.planes_t."""
        import torch
        if out is None:
            out = torch.empty((pr.n, 32), dtype=torch.int32, device=pr.lens.device)
        call("lrb_kmer_counts3t_dev", self._h, vp(pr.planes_t.data_ptr()),
             vp(pr.group_off.data_ptr()),
             vp(pr.order.data_ptr()) if pr.order is not None else None,
             vp(pr.lens.data_ptr()), pr.n, vp(out.data_ptr()))
        return out

    def make_codes_t(self, pr, sort=True):
        """Group-transposed 2-bit codes for the lane-per-read k=4 kernel, from pr.codes
        (pr.codes_t, pr.group_off4, pr.order4)."""
        import torch
        lens = pr.lens.cpu().numpy().view(np.uint32)[: pr.n].copy()
        goff = np.zeros((pr.n + 63) // 64 + 1, dtype=np.uint64)
        order = np.zeros(max(pr.n, 1), dtype=np.uint32) if sort else None
        call("lrb_codes_t_layout", _ptr(lens, u32p), pr.n, _ptr(order, u32p) if sort else None,
             _ptr(goff, u64p))
        dev = pr.lens.device
        pr.group_off4 = torch.from_numpy(goff.view(np.int64)).to(dev)
        pr.order4 = torch.from_numpy(order.view(np.int32)).to(dev) if sort else None
        pr.codes_t = torch.empty(max(int(goff[-1]) * 256, 8), dtype=torch.int32, device=dev)
        call("lrb_codes_t_from_codes_dev", self._h, vp(pr.codes.data_ptr()), vp(pr.code_off.data_ptr()),
             vp(pr.group_off4.data_ptr()), vp(pr.order4.data_ptr()) if sort else None, pr.n,
             vp(pr.codes_t.data_ptr()))
        return pr.codes_t

    def kmer_counts4t_dev(self, pr, out=None, k=4):
        """k=4 (or 5, or 3) tallies by the lane-per-read kernel on pr.codes_t."""
        import torch
        if out is None:
            out = torch.empty((pr.n, kmer_dim(k)), dtype=torch.int32, device=pr.lens.device)
        call("lrb_kmer_counts_t_dev", self._h, int(k), vp(pr.codes_t.data_ptr()), vp(pr.group_off4.data_ptr()),
             vp(pr.order4.data_ptr()) if pr.order4 is not None else None,
             vp(pr.lens.data_ptr()), pr.n, vp(out.data_ptr()))
        return out

    def kmer_counts3_dev(self, pr, mode=0, out=None):
        """k=3 tallies on the per-read layout (from the codes; the modes of earlier rounds are accepted and mean the same)."""
        import torch
        if out is None:
            out = torch.empty((pr.n, 32), dtype=torch.int32, device=pr.lens.device)
        pl = vp(pr.planes.data_ptr()) if pr.planes is not None else None
        mo = vp(pr.mask_off.data_ptr()) if pr.mask_off is not None else None
        co = vp(pr.code_off.data_ptr()) if pr.code_off is not None else None
        cd = vp(pr.codes.data_ptr()) if pr.codes is not None else None
        call("lrb_kmer_counts3_dev", self._h, cd, pl, co, mo, vp(pr.lens.data_ptr()), pr.n,
             int(mode), vp(out.data_ptr()))
        return out

    def k15_accumulate_dev(self, pr, table_t):
        call("lrb_k15_accumulate_dev", self._h, vp(pr.codes.data_ptr()), vp(pr.mask.data_ptr()),
             vp(pr.code_off.data_ptr()), vp(pr.mask_off.data_ptr()), vp(pr.lens.data_ptr()),
             pr.n, vp(table_t.data_ptr()))

    def k15_accumulate_part_dev(self, pr, table_t, max_windows=0):
        """k15_accumulate_dev under its older name (the partitioned forward route went in round 5)."""
        call("lrb_k15_accumulate_part_dev", self._h, vp(pr.codes.data_ptr()),
             vp(pr.mask.data_ptr()), vp(pr.code_off.data_ptr()), vp(pr.mask_off.data_ptr()),
             vp(pr.lens.data_ptr()), pr.n, int(max_windows), vp(table_t.data_ptr()))

    def k15_mirror_dev(self, table_t):
        call("lrb_k15_mirror_dev", self._h, vp(table_t.data_ptr()))

    # ---------------- the collective behind the C ABI (RCCL bound at run time) ---------------
    @staticmethod
    def rccl_unique_id():
        """128 bytes rank 0 hands to the other ranks (lrb_rccl_unique_id)."""
        buf = np.zeros(128, dtype=np.uint8)
        call("lrb_rccl_unique_id", _ptr(buf, u8p))
        return buf.tobytes()

    def rccl_comm_create(self, n_ranks, rank, uid):
        """ncclComm_t (as an integer handle) of this rank."""
        buf = np.frombuffer(bytes(uid), dtype=np.uint8).copy()
        h = vp()
        call("lrb_rccl_comm_create", self._h, int(n_ranks), int(rank), _ptr(buf, u8p), C.byref(h))
        return h.value

    @staticmethod
    def rccl_comm_destroy(comm):
        call("lrb_rccl_comm_destroy", vp(comm))

    def k15_allreduce(self, comm, buf_t):
        """In-place uint32 sum of a CUDA int32 tensor over the ranks of `comm`, on the context's stream."""
        call("lrb_k15_allreduce", self._h, vp(comm), vp(buf_t.data_ptr()), int(buf_t.numel()))
        return buf_t

    def k15_fold_half_dev(self, table_t, half_t=None):
        """Canonical half (2^29 uint32) of mirror(table) from the forward tallies (multi-GPU path)."""
        import torch
        if half_t is None:
            half_t = torch.empty(K15_HALF_ENTRIES, dtype=torch.int32, device=table_t.device)
        call("lrb_k15_fold_half_dev", self._h, vp(table_t.data_ptr()), vp(half_t.data_ptr()))
        return half_t

    def k15_expand_half_dev(self, half_t, table_t):
        """table[x] = table[rc(x)] = half[h(x)]."""
        call("lrb_k15_expand_half_dev", self._h, vp(half_t.data_ptr()), vp(table_t.data_ptr()))
        return table_t

    def cov_hist_dev(self, pr, table_t, bin_size, bins, hist=None, sums=None):
        import torch
        dev = pr.codes.device
        if hist is None:
            hist = torch.empty((pr.n, bins), dtype=torch.int32, device=dev)
        if sums is None:
            sums = torch.empty(max(pr.n, 1), dtype=torch.int32, device=dev)
        call("lrb_cov_hist_dev", self._h, vp(pr.codes.data_ptr()), vp(pr.mask.data_ptr()),
             vp(pr.code_off.data_ptr()), vp(pr.mask_off.data_ptr()), vp(pr.lens.data_ptr()),
             pr.n, vp(table_t.data_ptr()), int(bin_size), int(bins), vp(hist.data_ptr()),
             vp(sums.data_ptr()))
        return hist, sums[:pr.n]

    def cov_map_build_dev(self, table_t, bin_size, bins, map_t=None):
        """Compact map of the finished table for K3: one byte (bin id) per pair (x, rc(x)), 512 MB."""
        import torch
        if map_t is None:
            map_t = torch.empty(K15_HALF_ENTRIES, dtype=torch.uint8, device=table_t.device)
        call("lrb_cov_map_build_dev", self._h, vp(table_t.data_ptr()), int(bin_size), int(bins), vp(map_t.data_ptr()))
        return map_t

    def cov_hist_map_dev(self, pr, map_t, bins, hist=None, sums=None):
        """K3 against the compact map (same histograms as cov_hist_dev against the table)."""
        import torch
        dev = pr.codes.device
        if hist is None:
            hist = torch.empty((pr.n, bins), dtype=torch.int32, device=dev)
        if sums is None:
            sums = torch.empty(max(pr.n, 1), dtype=torch.int32, device=dev)
        call("lrb_cov_hist_map_dev", self._h, vp(pr.codes.data_ptr()), vp(pr.mask.data_ptr()),
             vp(pr.code_off.data_ptr()), vp(pr.mask_off.data_ptr()), vp(pr.lens.data_ptr()),
             pr.n, vp(map_t.data_ptr()), int(bins), vp(hist.data_ptr()), vp(sums.data_ptr()))
        return hist, sums[:pr.n]

    # ---------------- K2 and K3 on one partition of the windows (slice lists) ---------------
    def lists_geometry(self, n, bins=32, total_bases=0):
        """(reads per group, number of groups) of the window lists of n reads (total_bases long in all, 0 = unknown) for
        histograms of `bins` bins."""
        R, g = C.c_uint32(0), C.c_uint64(0)
        call("lrb_k15_lists_geometry_for", self._h, int(n), int(total_bases), int(bins), C.byref(R), C.byref(g))
        return R.value, g.value

    def lists_alloc(self, pr, bins=32):
        """Empty WindowLists for the resident reads `pr`: the list buffer (32 uint32 per mask word = 4 bytes per base
        slot), the bucket bounds of every group and the group bases."""
        import torch
        dev = pr.codes.device
        words = int((pr.mask_off[pr.n] - pr.mask_off[0]).item()) if pr.n else 0
        R, ngroups = self.lists_geometry(pr.n, bins, 32 * words)
        wl = WindowLists()
        wl.pr, wl.R, wl.ngroups, wl.bins = pr, R, ngroups, bins
        wl.n, wl.words = pr.n, words
        wl.lists = torch.empty(max(32 * words, 1) + 16, dtype=torch.int32, device=dev)
        wl.bounds = torch.empty(max(int(lib().lrb_k15_lists_bounds_words(ngroups)), 1), dtype=torch.int32, device=dev)
        wl.gbase = torch.empty(ngroups + 1, dtype=torch.int64, device=dev)
        return wl

    def lists_part_dev(self, pr, bins=32, out=None):
        """Window lists of the resident reads `pr` (lrb_k15_lists_part_dev) into `out` (lists_alloc of reads of the
        same count and mask words -- checked: the kernels write the whole of `lists` and `bounds`) or a new WindowLists."""
        wl = out if out is not None else self.lists_alloc(pr, bins)
        if out is not None:
            words = int((pr.mask_off[pr.n] - pr.mask_off[0]).item()) if pr.n else 0
            if out.n != pr.n or out.words < words or out.bins != bins:
                raise ValueError(f"lists_part_dev: `out` was allocated for {out.n} reads / {out.words} mask words / {out.bins} bins, "
                                 f"these are {pr.n} / {words} / {bins}")
            wl.pr = pr
        call("lrb_k15_lists_part_dev", self._h, vp(pr.codes.data_ptr()), vp(pr.mask.data_ptr()),
             vp(pr.code_off.data_ptr()), vp(pr.mask_off.data_ptr()), vp(pr.lens.data_ptr()), pr.n, wl.R,
             vp(wl.lists.data_ptr()), vp(wl.bounds.data_ptr()), vp(wl.gbase.data_ptr()))
        return wl

    def lists_tally_dev(self, wl, half_t):
        """K2 from the lists: half_t[h] += windows of wl's reads with pair index h (lrb_k15_lists_tally_dev)."""
        pr = wl.pr
        call("lrb_k15_lists_tally_dev", self._h, vp(pr.codes.data_ptr()), vp(pr.mask.data_ptr()),
             vp(pr.code_off.data_ptr()), vp(pr.mask_off.data_ptr()), vp(pr.lens.data_ptr()), pr.n, wl.R,
             vp(wl.lists.data_ptr()), vp(wl.bounds.data_ptr()), vp(wl.gbase.data_ptr()), vp(half_t.data_ptr()))
        return half_t

    def k15_accumulate_half_dev(self, pr, half_t):
        call("lrb_k15_accumulate_half_dev", self._h, vp(pr.codes.data_ptr()), vp(pr.mask.data_ptr()),
             vp(pr.code_off.data_ptr()), vp(pr.mask_off.data_ptr()), vp(pr.lens.data_ptr()), pr.n,
             vp(half_t.data_ptr()))
        return half_t

    def cov_map_build_half_dev(self, half_t, bin_size, bins, map_t=None):
        """The compact map from the canonical half of the table (same bytes as cov_map_build_dev on the mirrored table)."""
        import torch
        if map_t is None:
            map_t = torch.empty(K15_HALF_ENTRIES, dtype=torch.uint8, device=half_t.device)
        call("lrb_cov_map_build_half_dev", self._h, vp(half_t.data_ptr()), int(bin_size), int(bins), vp(map_t.data_ptr()))
        return map_t

    def cov_lists_sweep_dev(self, wl, map_t, bins, hist=None, sums=None):
        """K3 from lists a part call left: the sweep alone (lrb_cov_lists_sweep_dev).  Same histograms."""
        import torch
        pr = wl.pr
        if hist is None:
            hist = torch.empty((pr.n, int(bins)), dtype=torch.int32, device=pr.codes.device)
        if sums is None:
            sums = torch.empty(pr.n, dtype=torch.int32, device=pr.codes.device)
        call("lrb_cov_lists_sweep_dev", self._h, vp(pr.codes.data_ptr()), vp(pr.mask.data_ptr()),
             vp(pr.code_off.data_ptr()), vp(pr.mask_off.data_ptr()), vp(pr.lens.data_ptr()), pr.n, wl.R,
             vp(wl.lists.data_ptr()), vp(wl.bounds.data_ptr()), vp(wl.gbase.data_ptr()), vp(map_t.data_ptr()), int(bins),
             vp(hist.data_ptr()), vp(sums.data_ptr()))
        return hist, sums

    def cov_hist_sweep_dev(self, pr, map_t, bins, hist=None, sums=None):
        """K3 as a sweep over the compact map: the windows are partitioned by 2 MB map slice and every CU walks
        its read group's slice lists with the histograms in LDS (lrb_cov_hist_sweep_dev).  Same histograms."""
        import torch
        dev = pr.codes.device
        if hist is None:
            hist = torch.empty((pr.n, bins), dtype=torch.int32, device=dev)
        if sums is None:
            sums = torch.empty(max(pr.n, 1), dtype=torch.int32, device=dev)
        call("lrb_cov_hist_sweep_dev", self._h, vp(pr.codes.data_ptr()), vp(pr.mask.data_ptr()),
             vp(pr.code_off.data_ptr()), vp(pr.mask_off.data_ptr()), vp(pr.lens.data_ptr()),
             pr.n, vp(map_t.data_ptr()), int(bins), vp(hist.data_ptr()), vp(sums.data_ptr()))
        return hist, sums[:pr.n]

    def format_com_dev(self, counts_t, lens_t, k, want_q=True):
        """K8: com_profs text of device-resident counts -> (uint8 tensor [n * (9 dim + 1)], u32 q)."""
        import torch
        n, dim = counts_t.shape
        text = torch.empty(n * (9 * dim + 1) + 16, dtype=torch.uint8, device=counts_t.device)
        q = torch.empty((n, dim), dtype=torch.int32, device=counts_t.device) if want_q else None
        call("lrb_format_com_dev", self._h, vp(counts_t.data_ptr()), vp(lens_t.data_ptr()), n, dim, int(k),
             vp(text.data_ptr()), vp(q.data_ptr()) if want_q else None)
        text = text[: n * (9 * dim + 1)]
        return (text, q) if want_q else text

    def format_cov_dev(self, hist_t, sums_t, want_q=True):
        """K8: cov_profs text of device-resident histograms."""
        import torch
        n, bins = hist_t.shape
        text = torch.empty(n * 9 * bins + 16, dtype=torch.uint8, device=hist_t.device)
        q = torch.empty((n, bins), dtype=torch.int32, device=hist_t.device) if want_q else None
        call("lrb_format_cov_dev", self._h, vp(hist_t.data_ptr()), vp(sums_t.data_ptr()), n, bins,
             vp(text.data_ptr()), vp(q.data_ptr()) if want_q else None)
        text = text[: n * 9 * bins]
        return (text, q) if want_q else text

    def seed_dist_dev(self, M_t, seed, out=None):
        """0.5 - M @ M[seed] with out[seed] = 0 (calc_distances)."""
        import torch
        n, d = M_t.shape
        if out is None:
            out = torch.empty(n, dtype=torch.float32, device=M_t.device)
        call("lrb_seed_dist_dev", self._h, vp(M_t.data_ptr()), n, d, int(seed),
             vp(out.data_ptr()))
        return out

    def seed_hist_dev(self, M_t, seeds_t, out=None):
        """uint32 [S, 60]: histc(calc_distances(M, s), 60, 0, 0.3) for every seed."""
        import torch
        n, d = M_t.shape
        S = int(seeds_t.numel())
        if out is None:
            out = torch.empty((S, HIST_BINS), dtype=torch.int32, device=M_t.device)
        call("lrb_seed_hist_dev", self._h, vp(M_t.data_ptr()), n, d, vp(seeds_t.data_ptr()), S,
             vp(out.data_ptr()))
        return out


    def gauss_assign_dev(self, X_t, mean_t, std_t):
        """(best int32[U], best_p float64[U]): first-max cluster of the left-over likelihood."""
        import torch
        U, F = X_t.shape
        Cn = int(mean_t.shape[0])
        best = torch.empty(max(U, 1), dtype=torch.int32, device=X_t.device)
        bp = torch.empty(max(U, 1), dtype=torch.float64, device=X_t.device)
        call("lrb_gauss_assign_dev", self._h, vp(X_t.data_ptr()), U, F, vp(mean_t.data_ptr()),
             vp(std_t.data_ptr()), Cn, vp(best.data_ptr()), vp(bp.data_ptr()))
        return best[:U], bp[:U]


    # ---- K6: HDBSCAN (contigs pipeline) ------------------------------------------
    def hdb_core_dist_dev(self, X_t, k):
        """float32[n] distance of every row of X_t (cuda float32 n x dims) to its k-th nearest
        row, itself included."""
        import torch
        n, dims = X_t.shape
        X_t = X_t.contiguous()
        core = torch.empty(max(n, 1), dtype=torch.float32, device=X_t.device)
        call("lrb_hdb_core_dist_dev", self._h, vp(X_t.data_ptr()), n, dims, int(k), vp(core.data_ptr()))
        return core[:n]

    def hdb_mst_dev(self, X_t, core_t):
        """(u uint32[n-1], v uint32[n-1], w float32[n-1], rounds): minimum spanning tree under
        the mutual reachability distance; host arrays."""
        n, dims = X_t.shape
        X_t = X_t.contiguous()
        m = max(n - 1, 0)
        u, v = np.empty(m, np.uint32), np.empty(m, np.uint32)
        w = np.empty(m, np.float32)
        rounds = C.c_uint32(0)
        call("lrb_hdb_mst_dev", self._h, vp(X_t.data_ptr()), n, dims, vp(core_t.data_ptr()),
             u.ctypes.data_as(u32p), v.ctypes.data_as(u32p), w.ctypes.data_as(C.POINTER(C.c_float)),
             C.byref(rounds))
        return u, v, w, rounds.value

    def hdbscan(self, X, min_cluster_size=250, min_samples=None, core_excludes_self=None):
        """labels int32[n] of HDBSCAN(min_cluster_size, min_samples) on a host float32 matrix
        (hdbscan.HDBSCAN(...).fit_predict, cluster_utils.py:494).  -1 = noise.  core_excludes_self: the core distance
        goes to the min_samples-th OTHER point (True: the hdbscan package's Boruvka paths, which the reference's call
        takes) or to the min_samples-th with the point itself counted (False: sklearn.cluster.HDBSCAN, the package's
        Prim's paths); None = the library default (True unless LRB_HDB_CORE=self)."""
        X = np.ascontiguousarray(X, dtype=np.float32)
        n, dims = X.shape
        ms = int(min_cluster_size if min_samples is None else min_samples)
        labels = np.empty(max(n, 1), np.int32)
        nc = C.c_uint32(0)
        conv = -1 if core_excludes_self is None else int(bool(core_excludes_self))
        call("lrb_hdbscan_host_ex", self._h, X.ctypes.data_as(C.POINTER(C.c_float)), n, dims,
             int(min_cluster_size), ms, conv, labels.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(nc))
        return labels[:n]


class HostPacked:
    """A batch of reads in the packed HBM layout, in HOST memory: codes (2 bits a base), mask (1 bit a base), their word
    offsets per read, lengths, and the reads' byte offsets."""

    def __init__(self, codes, mask, code_off, mask_off, lens, offs):
        self.codes, self.mask, self.code_off, self.mask_off, self.lens, self.offs = codes, mask, code_off, mask_off, lens, offs
        self.n = len(offs) - 1
        self.total_bases = int(offs[-1] - offs[0])


def pack_reads_host(seqs, offs, scalar=False):
    """ASCII reads -> HostPacked on the host (lrb_pack_reads_host: what pack_kernel writes on the device, bit for bit;
    scalar=True: the loop for machines without AVX2 / BMI2)."""
    seqs, offs = _np(seqs, np.uint8), _np(offs, np.uint64)
    n = len(offs) - 1
    cw, mw = C.c_uint64(0), C.c_uint64(0)
    call("lrb_pack_host_sizes", _ptr(offs, u64p), n, C.byref(cw), C.byref(mw))
    codes, mask = np.empty(max(cw.value, 1), np.uint32), np.empty(max(mw.value, 1), np.uint32)
    co, mo = np.zeros(n + 1, np.uint64), np.zeros(n + 1, np.uint64)
    lens = np.zeros(max(n, 1), np.uint32)
    call("lrb_pack_reads_host_scalar" if scalar else "lrb_pack_reads_host", _ptr(seqs, u8p), _ptr(offs, u64p), n, _ptr(codes, u32p),
         _ptr(mask, u32p), _ptr(co, u64p), _ptr(mo, u64p), _ptr(lens, u32p))
    return HostPacked(codes[:cw.value], mask[:mw.value], co, mo, lens, offs)


def hdb_labels(n, u, v, w, min_cluster_size):
    """Host only: labels int32[n] from the n-1 spanning-tree edges (u, v, w)."""
    u = np.ascontiguousarray(u, np.uint32)
    v = np.ascontiguousarray(v, np.uint32)
    w = np.ascontiguousarray(w, np.float32)
    labels = np.empty(max(n, 1), np.int32)
    nc = C.c_uint32(0)
    call("lrb_hdb_labels", int(n), u.ctypes.data_as(u32p), v.ctypes.data_as(u32p),
         w.ctypes.data_as(C.POINTER(C.c_float)), int(min_cluster_size),
         labels.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(nc))
    return labels[:n], nc.value


def py_shuffle(ids):
    """``random.shuffle`` of an int64 array with the interpreter's own generator state (same
    draws, same result, state advanced as random.shuffle would) -- in the library."""
    import random
    x = np.ascontiguousarray(ids, dtype=np.int64).copy()
    version, internal, gauss = random.getstate()
    mt = np.array(internal[:624], dtype=np.uint32)
    pos = C.c_int(internal[624])
    call("lrb_mt_shuffle_i64", _ptr(mt, u32p), C.byref(pos), x.ctypes.data_as(C.POINTER(C.c_int64)), len(x))
    random.setstate((version, tuple(int(v) for v in mt) + (pos.value,), gauss))
    return x


# ---------------------------------------------------------------------------
# host-side helpers of the ABI (no GPU involved)
# ---------------------------------------------------------------------------
class FastxReader:
    """Batches of records from a FASTA/FASTQ(.gz) file (SeqReader semantics)."""

    def __init__(self, path):
        self._h = vp()
        call("lrb_reader_open", os.fsencode(path), C.byref(self._h))

    def next_batch(self, max_reads=10000, max_bytes=1 << 30):
        sp, op, n = u8p(), u64p(), C.c_uint64(0)
        call("lrb_reader_next", self._h, int(max_reads), int(max_bytes), C.byref(sp), C.byref(op),
             C.byref(n))
        n = n.value
        if n == 0:
            return None
        offs = np.ctypeslib.as_array(op, shape=(n + 1,)).copy()
        total = int(offs[-1])
        seqs = np.ctypeslib.as_array(sp, shape=(max(total, 1),)).copy()
        return seqs, offs

    def __iter__(self):
        while True:
            b = self.next_batch()
            if b is None:
                return
            yield b

    def close(self):
        if self._h:
            lib().lrb_reader_close(self._h)
            self._h = vp()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ParallelReader:
    """Batches of records from a pool of parser threads (lrb_preader_*).  Arrays returned
    by ``next_batch`` are views into library memory, valid until the next call."""

    def __init__(self, path, threads=8, chunk_bytes=1 << 28, rank=0, world=1, packed=False):
        """``rank``/``world``: hand out only byte ranges rank, rank+world, ... of the file.  ``packed``: the pool also packs
        every batch into the HBM layout in the thread that parsed it (next_packed)."""
        self._h = vp()
        self.packed = bool(packed)
        call("lrb_preader_open_ex", os.fsencode(path), int(threads), int(chunk_bytes), int(rank),
             int(world), 1 if packed else 0, C.byref(self._h))
        par, nr = C.c_int(0), C.c_uint64(0)
        call("lrb_preader_info", self._h, C.byref(par), C.byref(nr), None)
        self.parallel, self.n_ranges = bool(par.value), nr.value

    @property
    def last_range(self):
        """Index (in the whole file) of the byte range the last batch came from."""
        lr = C.c_uint64(0)
        call("lrb_preader_info", self._h, None, None, C.byref(lr))
        return lr.value

    def next_batch(self, copy=False):
        sp, op, n = u8p(), u64p(), C.c_uint64(0)
        call("lrb_preader_next", self._h, C.byref(sp), C.byref(op), C.byref(n))
        n = n.value
        if n == 0:
            return None
        offs = np.ctypeslib.as_array(op, shape=(n + 1,))
        seqs = np.ctypeslib.as_array(sp, shape=(max(int(offs[-1]), 1),))
        return (seqs.copy(), offs.copy()) if copy else (seqs, offs)

    def next_packed(self):
        """The next batch as a HostPacked (views into library memory, valid until the next call); None at the end.
        On a file the serial reader takes (gzip, FASTQ) the batch is packed here, in the calling thread."""
        b = self.next_batch(copy=False)
        if b is None:
            return None
        seqs, offs = b
        if not (self.packed and self.parallel):
            return pack_reads_host(seqs, offs)
        n = len(offs) - 1
        cp, mp, cop, mop, lp = u32p(), u32p(), u64p(), u64p(), u32p()
        call("lrb_preader_packed_view", self._h, C.byref(cp), C.byref(mp), C.byref(cop), C.byref(mop), C.byref(lp))
        co = np.ctypeslib.as_array(cop, shape=(n + 1,))
        mo = np.ctypeslib.as_array(mop, shape=(n + 1,))
        codes = np.ctypeslib.as_array(cp, shape=(max(int(co[-1]), 1),))
        mask = np.ctypeslib.as_array(mp, shape=(max(int(mo[-1]), 1),))
        lens = np.ctypeslib.as_array(lp, shape=(max(n, 1),))
        return HostPacked(codes, mask, co, mo, lens, offs)

    def __iter__(self):
        while True:
            b = self.next_batch(copy=True)
            if b is None:
                return
            yield b

    def close(self):
        if self._h:
            lib().lrb_preader_close(self._h)
            self._h = vp()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def read_all(path):
    """Whole file -> (uint8 buffer, uint64 offsets[n+1])."""
    bufs, offs, base = [], [np.zeros(1, np.uint64)], 0
    with FastxReader(path) as rd:
        for s, o in rd:
            bufs.append(s[: int(o[-1])])
            offs.append(o[1:] + np.uint64(base))
            base += int(o[-1])
    seqs = np.concatenate(bufs) if bufs else np.zeros(0, np.uint8)
    if seqs.size == 0:
        seqs = np.zeros(1, np.uint8)
    return seqs, np.concatenate(offs)


def format_com(counts, lens, k, threads=1, want_values=False):
    """com_profs text (bytes) [+ the float64 values that text parses back to]."""
    counts = _np(counts, np.uint32)
    lens = _np(lens, np.uint32)
    n, dim = counts.shape
    buf = C.create_string_buffer(int(lib().lrb_profile_text_bound(n, dim)))
    w = C.c_uint64(0)
    vals = np.zeros((n, dim), dtype=np.float64) if want_values else None
    call("lrb_format_com", _ptr(counts, u32p), _ptr(lens, u32p), n, dim, int(k), int(threads),
         buf, C.byref(w), _ptr(vals, f64p) if want_values else None)
    txt = buf.raw[: w.value]
    return (txt, vals) if want_values else txt


def format_cov(hist, sums, threads=1, want_values=False):
    """cov_profs text (bytes) [+ parsed values]."""
    hist = _np(hist, np.uint32)
    sums = _np(sums, np.uint32)
    n, bins = hist.shape
    buf = C.create_string_buffer(int(lib().lrb_profile_text_bound(n, bins)))
    w = C.c_uint64(0)
    vals = np.zeros((n, bins), dtype=np.float64) if want_values else None
    call("lrb_format_cov", _ptr(hist, u32p), _ptr(sums, u32p), n, bins, int(threads), buf,
         C.byref(w), _ptr(vals, f64p) if want_values else None)
    txt = buf.raw[: w.value]
    return (txt, vals) if want_values else txt
