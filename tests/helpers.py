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
