"""Shared helpers for the test-suite (test infrastructure)."""
import gzip
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def gz_bytes(name):
    with gzip.open(os.path.join(GOLDEN, name), "rb") as f:
        return f.read()


def golden_path(name):
    return os.path.join(GOLDEN, name)


def parse_profile_text(txt):
    rows = [list(map(float, ln.split())) for ln in txt.decode().split("\n") if ln.strip()]
    return np.array(rows, dtype=np.float64)


def random_reads(rng, n, lo, hi, p_n=0.0, p_lower=0.0):
    """n random reads with lengths in [lo, hi]; optional N / lowercase noise."""
    out = []
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    for _ in range(n):
        L = int(rng.integers(lo, hi + 1))
        s = rng.choice(alpha, size=L)
        if p_n > 0 and L:
            m = rng.random(L) < p_n
            s = np.where(m, ord("N"), s).astype(np.uint8)
        if p_lower > 0 and L:
            m = rng.random(L) < p_lower
            s = np.where(m, s | 0x20, s).astype(np.uint8)
        out.append(bytes(s))
    return out


# ---------------------------------------------------------------------------
# numpy model of the packed HBM layout (include/lrb_hip.h) -- used to check the
# pack kernel and to turn device-generated packed reads back into ASCII.
# ---------------------------------------------------------------------------
def np_pack(buf, offs):
    """-> (codes u32[], mask u32[], code_off, mask_off, lens) exactly as lrb_pack_reads_dev."""
    n = len(offs) - 1
    lens = np.diff(offs).astype(np.int64)
    cw = ((-(-lens // 16) + 3) // 4) * 4 + 4
    mw = ((-(-lens // 32) + 3) // 4) * 4 + 4
    co = np.zeros(n + 1, np.uint64)
    mo = np.zeros(n + 1, np.uint64)
    co[1:] = np.cumsum(cw)
    mo[1:] = np.cumsum(mw)
    codes = np.zeros(int(co[-1]), np.uint32)
    mask = np.zeros(int(mo[-1]), np.uint32)
    for r in range(n):
        s = buf[int(offs[r]):int(offs[r + 1])].astype(np.uint32)
        L = len(s)
        if L == 0:
            continue
        code = (s >> 1) & 3
        i = np.arange(L)
        np.bitwise_or.at(codes, int(co[r]) + i // 16, (code << (30 - 2 * (i % 16))).astype(np.uint32))
        ok = ((s == 65) | (s == 67) | (s == 71) | (s == 84)).astype(np.uint32)
        np.bitwise_or.at(mask, int(mo[r]) + i // 32, (ok << (31 - (i % 32))).astype(np.uint32))
    return codes, mask, co, mo, lens.astype(np.uint32)


def np_unpack(codes_words, L):
    """Packed words of one read -> the ASCII read whose every byte is one of ACTG."""
    w = np.asarray(codes_words, dtype=np.uint32)
    i = np.arange(L)
    code = (w[i // 16] >> (30 - 2 * (i % 16)).astype(np.uint32)) & 3
    return np.frombuffer(b"ACTG", dtype=np.uint8)[code]
