"""ctypes binding of the CPU oracle (oracle/lrb_oracle.c) -- TEST INFRASTRUCTURE.

Only tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module, and only as the checker.  The product
package (``lrbinner_amd``) never imports it.

Each wrapper names the C function it calls; the C function cites the reference
file:line it restates.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblrb_oracle.so")
_lib = None

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
f64p = C.POINTER(C.c_double)


def build():
    """Compile liblrb_oracle.so (and oracle/_ref when the reference sources exist)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "lrb_oracle.c")
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", _HERE, "liblrb_oracle.so"])
        L = C.CDLL(_SO)
        L.orc_revcomp.restype = C.c_uint64
        L.orc_revcomp.argtypes = [C.c_uint64, C.c_uint]
        L.orc_kmer_lut.restype = C.c_uint32
        L.orc_kmer_lut.argtypes = [C.c_uint, u32p]
        L.orc_count_kmers_batch.restype = None
        L.orc_count_kmers_batch.argtypes = [u8p, u64p, C.c_uint64, C.c_uint, u32p, u64p]
        L.orc_com_profile.restype = None
        L.orc_com_profile.argtypes = [u32p, C.c_uint32, C.c_uint64, f64p]
        L.orc_k15_max_windows.restype = C.c_uint64
        L.orc_k15_max_windows.argtypes = [u64p, C.c_uint64]
        L.orc_k15_sparse.restype = C.c_uint64
        L.orc_k15_sparse.argtypes = [u8p, u64p, C.c_uint64, u32p, u32p]
        L.orc_k15_accumulate_dense.restype = None
        L.orc_k15_accumulate_dense.argtypes = [u8p, C.c_uint64, u32p]
        L.orc_cov_bin.restype = C.c_long
        L.orc_cov_bin.argtypes = [C.c_long, C.c_long, C.c_int]
        L.orc_cov_hist_batch.restype = None
        L.orc_cov_hist_batch.argtypes = [u8p, u64p, C.c_uint64, u32p, u32p, u32p, C.c_uint64,
                                         C.c_long, C.c_int, u32p, u64p]
        L.orc_cov_profile.restype = None
        L.orc_cov_profile.argtypes = [u32p, C.c_int, C.c_uint64, f64p]
        L.orc_format_com_row.restype = C.c_uint64
        L.orc_format_com_row.argtypes = [f64p, C.c_uint32, C.c_char_p]
        L.orc_format_cov_row.restype = C.c_uint64
        L.orc_format_cov_row.argtypes = [f64p, C.c_uint32, C.c_char_p]
        L.orc_fastx_read.restype = C.c_int
        L.orc_fastx_read.argtypes = [C.c_char_p, C.POINTER(u8p), C.POINTER(u64p), u64p]
        L.orc_synth_markov.restype = None
        L.orc_synth_markov.argtypes = [C.c_uint64, C.c_uint, f64p, C.c_uint64, u8p]
        L.orc_synth_read.restype = C.c_uint64
        L.orc_synth_read.argtypes = [C.c_uint64, u8p, C.c_uint64, C.c_double, C.c_double, C.c_double,
                                     C.c_int, u8p]
        L.orc_free.restype = None
        L.orc_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def ref_bin(name):
    """Path of a real reference binary in oracle/_ref, or None when absent."""
    p = os.path.join(_HERE, "_ref", name)
    return p if os.path.exists(p) else None


def _p(a, t):
    return a.ctypes.data_as(t)


def concat(reads):
    """list[bytes] -> (uint8 buffer, uint64 offsets[n+1])."""
    offs = np.zeros(len(reads) + 1, dtype=np.uint64)
    if reads:
        offs[1:] = np.cumsum([len(r) for r in reads], dtype=np.uint64)
    buf = np.frombuffer(b"".join(reads), dtype=np.uint8).copy() if reads else np.zeros(0, np.uint8)
    if buf.size == 0:
        buf = np.zeros(1, np.uint8)
    return buf, offs


def revcomp(x, k):
    return int(lib().orc_revcomp(int(x), int(k)))


def kmer_lut(k):
    lut = np.zeros(4 ** k, dtype=np.uint32)
    D = lib().orc_kmer_lut(k, _p(lut, u32p))
    return lut, int(D)


def kmer_dim(k):
    return kmer_lut(k)[1]


def count_kmers(buf, offs, k):
    """-> (counts uint32[n, D], totals uint64[n])   [orc_count_kmers_batch]"""
    n = len(offs) - 1
    D = kmer_dim(k)
    counts = np.zeros((n, D), dtype=np.uint32)
    totals = np.zeros(n, dtype=np.uint64)
    if n:
        lib().orc_count_kmers_batch(_p(buf, u8p), _p(offs, u64p), n, k, _p(counts, u32p),
                                    _p(totals, u64p))
    return counts, totals


def com_profile(counts, totals):
    """-> float64[n, D]   [orc_com_profile]"""
    n, D = counts.shape
    out = np.zeros((n, D), dtype=np.float64)
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    for r in range(n):
        lib().orc_com_profile(_p(counts[r], u32p), D, int(totals[r]), _p(out[r], f64p))
    return out


def k15_sparse(buf, offs):
    """-> (keys uint32[u] sorted unique, counts uint32[u])   [orc_k15_sparse]"""
    n = len(offs) - 1
    m = int(lib().orc_k15_max_windows(_p(offs, u64p), n))
    keys = np.zeros(max(2 * m, 1), dtype=np.uint32)
    cnts = np.zeros(max(2 * m, 1), dtype=np.uint32)
    u = int(lib().orc_k15_sparse(_p(buf, u8p), _p(offs, u64p), n, _p(keys, u32p), _p(cnts, u32p)))
    return keys[:u].copy(), cnts[:u].copy()


def cov_bin(count, bin_size, bins):
    return int(lib().orc_cov_bin(int(count), int(bin_size), int(bins)))


def cov_hist(buf, offs, keys, cnts, bin_size, bins, table_dense=None):
    """-> (hist uint32[n, bins], sums uint64[n])   [orc_cov_hist_batch]"""
    n = len(offs) - 1
    hist = np.zeros((n, bins), dtype=np.uint32)
    sums = np.zeros(n, dtype=np.uint64)
    if n:
        td = _p(table_dense, u32p) if table_dense is not None else None
        kk = _p(keys, u32p) if keys is not None and len(keys) else None
        cc = _p(cnts, u32p) if cnts is not None and len(cnts) else None
        nu = 0 if keys is None else len(keys)
        lib().orc_cov_hist_batch(_p(buf, u8p), _p(offs, u64p), n, td, kk, cc, nu,
                                 int(bin_size), int(bins), _p(hist, u32p), _p(sums, u64p))
    return hist, sums


def cov_profile(hist, sums):
    """-> float64[n, bins]   [orc_cov_profile]"""
    n, bins = hist.shape
    out = np.zeros((n, bins), dtype=np.float64)
    hist = np.ascontiguousarray(hist, dtype=np.uint32)
    for r in range(n):
        lib().orc_cov_profile(_p(hist[r], u32p), bins, int(sums[r]), _p(out[r], f64p))
    return out


def format_com(prof):
    """float64[n, D] -> bytes of the com_profs text   [orc_format_com_row]"""
    n, D = prof.shape
    b = C.create_string_buffer(25 * (D + 1))
    out = []
    prof = np.ascontiguousarray(prof, dtype=np.float64)
    for r in range(n):
        l = lib().orc_format_com_row(_p(prof[r], f64p), D, b)
        out.append(b.raw[:l])
    return b"".join(out)


def format_cov(prof):
    """float64[n, bins] -> bytes of the cov_profs text   [orc_format_cov_row]"""
    n, D = prof.shape
    b = C.create_string_buffer(25 * (D + 1))
    out = []
    prof = np.ascontiguousarray(prof, dtype=np.float64)
    for r in range(n):
        l = lib().orc_format_cov_row(_p(prof[r], f64p), D, b)
        out.append(b.raw[:l])
    return b"".join(out)


def fastx_read(path):
    """-> (uint8 buffer, uint64 offsets[n+1]); raises OSError   [orc_fastx_read]"""
    sp, op, n = u8p(), u64p(), C.c_uint64(0)
    rc = lib().orc_fastx_read(os.fsencode(path), C.byref(sp), C.byref(op), C.byref(n))
    if rc != 0:
        raise OSError(f"cannot open {path}")
    n = n.value
    offs = np.ctypeslib.as_array(op, shape=(n + 1,)).copy()
    total = int(offs[-1])
    buf = np.ctypeslib.as_array(sp, shape=(max(total, 1),)).copy()
    lib().orc_free(sp)
    lib().orc_free(op)
    return buf, offs


def reads_of(buf, offs):
    return [bytes(buf[int(offs[i]):int(offs[i + 1])]) for i in range(len(offs) - 1)]


def synth_markov(seed, order, cum, length):
    """Order-`order` Markov genome as ASCII uint8[length]   [orc_synth_markov]"""
    cum = np.ascontiguousarray(cum, dtype=np.float64)
    assert cum.shape == (4 ** order, 4)
    out = np.empty(length, dtype=np.uint8)
    lib().orc_synth_markov(int(seed), int(order), _p(cum, f64p), int(length), _p(out, u8p))
    return out


def synth_read(seed, src, p_sub, p_del, p_ins, rc):
    """Noisy copy of `src` (uint8 ASCII) -> bytes   [orc_synth_read]"""
    src = np.ascontiguousarray(src, dtype=np.uint8)
    dst = np.empty(2 * len(src) + 4, dtype=np.uint8)
    m = lib().orc_synth_read(int(seed), _p(src, u8p), len(src), p_sub, p_del, p_ins, int(rc), _p(dst, u8p))
    return dst[:m].tobytes()
