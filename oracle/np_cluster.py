"""numpy restatement of the reference's clustering arithmetic -- TEST INFRASTRUCTURE.

Plain, sequential restatement of ``mbcclr_utils/cluster_utils.py`` (each function
cites the lines it follows), pinned by tests/test_oracle_py_golden.py against
vectors produced by importing the reference itself
(tests/golden/make_golden_py.py -> py_cluster.npz).  Only tests/, smoke() and the
cpu_baseline leg of bench.py may import this module.
"""
import math
import random
from collections import defaultdict

import numpy as np

_DELTA_X = 0.005  # cluster_utils.py:52
_XMAX = 0.3       # cluster_utils.py:53
NBINS = math.ceil(_XMAX / _DELTA_X)  # 60


def normal_pdf_kernel():
    """31 samples of N(0, 0.01) on [-0.075, 0.075] times _DELTA_X, in float32
    (cluster_utils.py:58-67 tabulates the same values to 9 digits)."""
    x = (np.arange(31) - 15) * _DELTA_X
    pdf = np.exp(-0.5 * (x / 0.01) ** 2) / (0.01 * math.sqrt(2 * math.pi))
    # the reference holds the table as 9-significant-digit decimals
    pdf = np.array([float(f"{v:.8e}") for v in pdf])
    return (pdf.astype(np.float32) * np.float32(_DELTA_X)).astype(np.float32)


_PDF = normal_pdf_kernel()


def normalize(matrix):
    """cluster_utils.py:31-42: zero rows -> 1/ncols; rows /= (||row|| * sqrt 2)."""
    m = np.array(matrix, dtype=np.float32, copy=True)
    zero = m.sum(axis=1, dtype=np.float32) == 0
    m[zero] = np.float32(1 / m.shape[1])
    nrm = np.sqrt((m * m).sum(axis=1, dtype=np.float32)).astype(np.float32)
    m /= (nrm.reshape(-1, 1) * np.float32(2 ** 0.5))
    return m


def calc_distances(matrix, index):
    """cluster_utils.py:45-49.  float32 dot as an ordered fma-free loop is not what
    BLAS does either; parity with the reference is within 1e-6 absolute."""
    d = (np.float32(0.5) - matrix @ matrix[index]).astype(np.float32)
    d[index] = 0.0
    return d


def histc(x, bins=NBINS, lo=0.0, hi=_XMAX):
    """torch.histc(x, 60, 0, 0.3) on float32 input (cluster_utils.py:138): position
    = (x - lo) * bins / (hi - lo) in float32, truncated; hi itself goes to the last
    bin; values outside [lo, hi] are dropped.  Returns float32 counts."""
    x = np.asarray(x, dtype=np.float32)
    lo32, hi32 = np.float32(lo), np.float32(hi)
    keep = (x >= lo32) & (x <= hi32)
    v = x[keep]
    pos = (((v - lo32) * np.float32(bins)) / (hi32 - lo32)).astype(np.int64)
    pos[pos == bins] = bins - 1
    return np.bincount(pos, minlength=bins).astype(np.float32)


def calc_densities(histogram):
    """cluster_utils.py:69-82: hist (*) 31-tap kernel, accumulated bin by bin in
    float32, cropped [15:-15]."""
    h = np.asarray(histogram, dtype=np.float32)
    dens = np.zeros(len(h) + len(_PDF) - 1, dtype=np.float32)
    for i in range(len(h)):
        dens[i:i + len(_PDF)] += _PDF * h[i]
    return dens[15:-15]


def find_valley_ratio(densities):
    """cluster_utils.py:87-133.  Returns (ratio, maxima, early_minima, minima) or four
    False.  x accumulates in float64; density arithmetic is float32."""
    d = np.asarray(densities, dtype=np.float32)
    peak_density = np.float32(0)
    min_density = None
    peak_over = False
    minima = maxima = early_minima = None
    x = 0
    with np.errstate(all="ignore"):
        for n in range(len(d)):
            density = d[n]
            if not peak_over and density > peak_density:
                if x > 0.1:
                    break
                peak_density = density
                maxima = x
            if not peak_over and density < peak_density:
                peak_over = True
                peak_density = density  # the peak is overwritten by the first lower value
                min_density = density
                minima = x
            if peak_over and density > min_density:
                break
            if peak_over and density < min_density:
                min_density = density
                minima = x
                drop = (d[n - 1] - d[n]) / np.float32(1 / _DELTA_X)
                if n != 0 and drop > 0.5:
                    early_minima = x
                if drop < 0.2:
                    break
            x += _DELTA_X
        if not peak_over:
            return False, False, False, False
        if early_minima is None:
            early_minima = minima
        return min_density / peak_density, maxima, early_minima, minima


def _seed_profile(matrix, seed):
    distances = calc_distances(matrix, seed)
    histogram = histc(distances)
    histogram[0] -= 1
    return distances, find_valley_ratio(calc_densities(histogram))


def get_cluster_center(matrix, seed):
    """cluster_utils.py:136-192."""
    distances, (ratio, chosen_peak, chosen_minima, chosen_tail) = _seed_profile(matrix, seed)
    with np.errstate(all="ignore"):
        if not chosen_peak or ratio > 0.5:
            return False, False, False, False, False
    from_x, to_x = chosen_peak - _DELTA_X * 5, chosen_peak + _DELTA_X * 5
    chosen_points = np.flatnonzero((distances > from_x) & (distances < to_x)).tolist()
    if len(chosen_points) < 100:
        return False, False, False, False, False
    sample_size = int(min(1000, max(100, len(chosen_points) * 0.01)))
    sampled_points = random.sample(chosen_points, sample_size)
    ratio = 10000
    best_point = distance_cache = tail = minima = maxima = None
    for p in sampled_points:
        distances, (new_ratio, new_maxima, new_minima, new_tail) = _seed_profile(matrix, p)
        with np.errstate(all="ignore"):
            if new_ratio and new_ratio < ratio:
                ratio, best_point, distance_cache = new_ratio, p, distances
                tail, minima, maxima = new_tail, new_minima, new_maxima
    return best_point, distance_cache, maxima, minima, tail


def cluster_points(latent, iterations, min_cluster_size):
    """cluster_utils.py:195-258.  Returns {cluster id: set(read index)}."""
    matrix = normalize(latent)
    clusters = defaultdict(list)
    read_ids = np.arange(len(matrix))
    read_ids_ref = np.arange(len(matrix))

    def peel(x, distance_cache, tail):
        nonlocal matrix, read_ids, read_ids_ref
        removables = np.flatnonzero(distance_cache <= tail)
        clusters[x] = set(read_ids_ref[removables])
        keep = np.ones(len(read_ids_ref), dtype=bool)
        keep[removables] = False
        read_ids_ref = read_ids_ref[keep]
        matrix = matrix[keep]
        read_ids = np.arange(len(read_ids_ref))

    if iterations != 0:
        for x in range(iterations):
            if len(read_ids) < min_cluster_size * 0.6:
                break
            random_point = random.choice(read_ids)
            _, distance_cache, _, _, tail = get_cluster_center(matrix, random_point)
            if tail:
                peel(x, distance_cache, tail)
    else:
        x = 0
        while True:
            if len(read_ids) < min_cluster_size * 0.1:
                break
            finish_search = True
            random_candidates = list(read_ids)
            random.shuffle(random_candidates)
            for random_point in random_candidates:
                _, distance_cache, _, _, tail = get_cluster_center(matrix, random_point)
                if tail:
                    peel(x, distance_cache, tail)
                    x += 1
                    finish_search = False
                    break
            if finish_search:
                break
    return clusters


def normal(val, mean, std):
    """cluster_utils.py:261-268: sum of log(N(val; mean, std) + 1e-7) (nan when a std is 0)."""
    with np.errstate(all="ignore"):
        a = np.sqrt(2 * np.pi) * std
        b = np.exp(-0.5 * np.square((val - mean) / std))
        return np.sum(np.log(b / a + 0.0000001))
