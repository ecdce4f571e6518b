"""Clustering stage -- drop-in for the reads path of ``mbcclr_utils/cluster_utils.py``.

Same algorithm, constants, random-number call sequence and output files as the
reference (cluster_utils.py:23-362): seeded cosine-distance histograms -> smoothed
density -> valley -> peel the cluster off; left-over reads go to the cluster with
the highest diagonal-Gaussian likelihood.

What moved to the GPU (liblrb_hip.so, through ``lrbinner_amd.device``):
  * calc_distances        -> lrb_seed_dist_dev   (one matvec, result stays in HBM)
  * the 1 + <=1000 histogram passes of get_cluster_center
                          -> lrb_seed_hist_dev   (all seeds in ONE pass over M)
  * np.delete on the matrix -> boolean compaction in HBM
  * the left-over likelihoods -> lrb_gauss_assign_dev (float64, one wave per read)
The 60-bin density / valley logic is a few hundred flops and stays on the host,
bit-for-bit as in the reference (float32 densities, float64 x accumulation).
"""
import logging
import math
import os
import pickle
import random
import shutil
from collections import defaultdict

import numpy as np

from . import _npcache

logger = logging.getLogger('LRBinner')

_DELTA_X = 0.005
_XMAX = 0.3
_NBINS = math.ceil(_XMAX / _DELTA_X)
# candidates whose phase-1 histograms are computed per launch in the exhaustive sweep
_PREFETCH = 256


def _normal_pdf_kernel():
    """N(0, 0.01) sampled at 31 points on [-0.075, 0.075], times _DELTA_X, float32 --
    the table of cluster_utils.py:58-67 (which lists it to 9 significant digits)."""
    x = (np.arange(31) - 15) * _DELTA_X
    pdf = np.exp(-0.5 * (x / 0.01) ** 2) / (0.01 * math.sqrt(2 * math.pi))
    pdf = np.array([float(f"{v:.8e}") for v in pdf])
    return (pdf.astype(np.float32) * np.float32(_DELTA_X)).astype(np.float32)


_NORMALPDF = _normal_pdf_kernel()


def calc_densities(histogram, pdf=_NORMALPDF):
    """Histogram smoothed with the 31-tap kernel; float32 accumulation bin by bin,
    "same" crop (cluster_utils.py:69-82)."""
    h = np.asarray(histogram, dtype=np.float32)
    dens = np.zeros(len(h) + len(pdf) - 1, dtype=np.float32)
    for i in range(len(h)):
        dens[i:i + len(pdf)] += pdf * h[i]
    return dens[15:-15]


def find_valley_ratio(densities):
    """(valley/peak ratio, maxima, early_minima, minima) or four False (the contract of
    cluster_utils.py:87-133) for ONE density row: row 0 of the vectorised scan below."""
    ok, ratio, maxima, early, minima = find_valley_ratio_batch(np.asarray(densities, dtype=np.float32)[None, :])
    if not ok[0]:
        return False, False, False, False
    none = lambda v: None if np.isnan(v) else float(v)
    return ratio[0], none(maxima[0]), none(early[0]), none(minima[0])


def calc_densities_batch(histograms, pdf=_NORMALPDF):
    """calc_densities for S histograms at once: the same float32 additions in the same order
    per element (bin by bin), vectorised over the histograms."""
    h = np.asarray(histograms, dtype=np.float32)
    dens = np.zeros((h.shape[0], h.shape[1] + len(pdf) - 1), dtype=np.float32)
    for i in range(h.shape[1]):
        dens[:, i:i + len(pdf)] += pdf[None, :] * h[:, i:i + 1]
    return dens[:, 15:-15]


def find_valley_ratio_batch(densities):
    """find_valley_ratio for S density rows at once.  Returns (valid, ratio, maxima,
    early_minima, minima): ``valid`` is False where the scalar version returns four False;
    the other arrays hold exactly the scalar version's values where valid."""
    d = np.asarray(densities, dtype=np.float32)
    S, n_bins = d.shape
    peak = np.zeros(S, np.float32)
    mind = np.zeros(S, np.float32)
    peak_over = np.zeros(S, bool)
    active = np.ones(S, bool)
    maxima = np.full(S, np.nan)
    minima = np.full(S, np.nan)
    early = np.full(S, np.nan)
    x = 0  # the same float64 accumulation as the scalar loop
    with np.errstate(all="ignore"):
        for n in range(n_bins):
            dn = d[:, n]
            rising = active & ~peak_over & (dn > peak)
            if x > 0.1:
                active &= ~rising           # a new peak this far out: stop, state unchanged
                rising = np.zeros(S, bool)
            peak = np.where(rising, dn, peak)
            maxima = np.where(rising, x, maxima)
            falls = active & ~peak_over & (dn < peak)
            peak_over |= falls
            peak = np.where(falls, dn, peak)
            mind = np.where(falls, dn, mind)
            minima = np.where(falls, x, minima)
            up = active & peak_over & ~falls & (dn > mind)
            active &= ~up
            down = active & peak_over & ~falls & (dn < mind)
            mind = np.where(down, dn, mind)
            minima = np.where(down, x, minima)
            if n != 0:
                drop = (d[:, n - 1] - dn) / np.float32(1 / _DELTA_X)
                early = np.where(down & (drop > 0.5), x, early)
                active &= ~(down & (drop < 0.2))
            else:
                # drop uses d[-1] in the scalar loop; only its "< 0.2 -> stop" branch applies at n = 0
                drop = (d[:, -1] - dn) / np.float32(1 / _DELTA_X)
                active &= ~(down & (drop < 0.2))
            x += _DELTA_X
        early = np.where(np.isnan(early), minima, early)
        ratio = mind / peak
    return peak_over, ratio, maxima, early, minima


def _valley_of_hist(hist_counts):
    h = np.asarray(hist_counts).astype(np.float32)
    h[0] -= 1  # the seed itself (cluster_utils.py:139)
    return find_valley_ratio(calc_densities(h))


def _valleys_of_hists(hists):
    """_valley_of_hist for a block of candidates in one vectorised scan (the same float32 operations per row):
    the exhaustive search looks at hundreds to thousands of candidates per run, and a scan of ONE row costs what a
    scan of 256 does."""
    h = np.asarray(hists).astype(np.float32)
    if h.shape[0] == 0:
        return []
    h[:, 0] -= 1
    ok, ratio, maxima, early, minima = find_valley_ratio_batch(calc_densities_batch(h))
    none = lambda v: None if np.isnan(v) else float(v)
    return [(ratio[i], none(maxima[i]), none(early[i]), none(minima[i])) if ok[i] else (False, False, False, False)
            for i in range(h.shape[0])]


class HipBackend:
    """The normalised latent matrix resident in HBM + the two K4 kernels."""

    def __init__(self, device_index=None):
        import torch
        from . import device as lrb
        if device_index is None:
            device_index = int(os.environ.get("LRB_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        self.torch = torch
        self.dev = torch.device("cuda", device_index)
        torch.cuda.set_device(self.dev)
        self.ctx = lrb.Context(device_index, use_torch_stream=True)
        self.M = None

    def load(self, latent):
        """normalize() of cluster_utils.py:31-42 on the device."""
        t = self.torch
        m = t.as_tensor(np.asarray(latent), dtype=t.float32).to(self.dev).clone()
        zeromask = m.sum(dim=1) == 0
        m[zeromask] = 1 / m.shape[1]
        m /= (m.norm(dim=1).reshape(-1, 1) * (2 ** 0.5))
        self.M = m.contiguous()

    def __len__(self):
        return int(self.M.shape[0])

    def distances(self, idx):
        d = self.ctx.seed_dist_dev(self.M, int(idx))
        return d.cpu().numpy()

    def seed_hists(self, seeds):
        t = self.torch
        s = t.as_tensor(np.asarray(seeds, dtype=np.int64)).to(self.dev)
        h = self.ctx.seed_hist_dev(self.M, s)
        return h.cpu().numpy().view(np.uint32)

    def remove(self, removables):
        t = self.torch
        keep = t.ones(len(self), dtype=t.bool, device=self.dev)
        keep[t.as_tensor(np.asarray(removables, dtype=np.int64)).to(self.dev)] = False
        self.M = self.M[keep].contiguous()


def get_cluster_center(backend, seed, seed_hist=None, valley=None):
    """cluster_utils.py:136-192 with the histogram passes batched.  ``valley``: the seed histogram's valley scan
    when the caller has done it for a block of candidates already."""
    if valley is None:
        if seed_hist is None:
            seed_hist = backend.seed_hists([seed])[0]
        valley = _valley_of_hist(seed_hist)
    ratio, chosen_peak, chosen_minima, chosen_tail = valley
    with np.errstate(all="ignore"):
        if not chosen_peak or ratio > 0.5:
            return False, False, False, False, False
    distances = backend.distances(seed)
    from_x, to_x = chosen_peak - _DELTA_X * 5, chosen_peak + _DELTA_X * 5
    chosen_points = np.flatnonzero((distances > from_x) & (distances < to_x)).tolist()
    if len(chosen_points) < 100:
        return False, False, False, False, False
    sample_size = int(min(1000, max(100, len(chosen_points) * 0.01)))
    sampled_points = random.sample(chosen_points, sample_size)

    hists = backend.seed_hists(sampled_points)  # one pass over M for all samples
    h = np.asarray(hists).astype(np.float32)
    h[:, 0] -= 1  # the seeds themselves (cluster_utils.py:139)
    valid, ratios, maxs, earlies, tails = find_valley_ratio_batch(calc_densities_batch(h))
    # the sequential scan keeps the FIRST sample with the smallest truthy ratio below 10000
    with np.errstate(all="ignore"):
        ok = valid & (ratios != 0) & (ratios < 10000)
    best_point = tail = minima = maxima = None
    if ok.any():
        cand = np.where(ok, ratios, np.float32(np.inf))
        b = int(np.argmin(cand))
        best_point = sampled_points[b]
        tail, minima, maxima = tails[b], earlies[b], maxs[b]
    distance_cache = backend.distances(best_point) if best_point is not None else None
    return best_point, distance_cache, maxima, minima, tail


def cluster_points(latent, iterations, min_cluster_size, backend=None):
    """cluster_utils.py:195-258.  Returns {cluster id: set(read index)}."""
    if backend is None:
        backend = HipBackend()
    backend.load(latent)
    clusters = defaultdict(list)
    read_ids = np.arange(len(backend))
    read_ids_ref = np.arange(len(backend))

    def peel(x, distance_cache, tail):
        nonlocal read_ids, read_ids_ref
        removables = np.flatnonzero(distance_cache <= tail)
        clusters[x] = set(read_ids_ref[removables])
        keep = np.ones(len(read_ids_ref), dtype=bool)
        keep[removables] = False
        read_ids_ref = read_ids_ref[keep]
        backend.remove(removables)
        read_ids = np.arange(len(read_ids_ref))

    if iterations != 0:
        for x in range(iterations):
            if len(read_ids) < min_cluster_size * 0.6:
                break
            random_point = random.choice(read_ids)
            _, distance_cache, _, _, tail = get_cluster_center(backend, random_point)
            if tail:
                peel(x, distance_cache, tail)
    else:
        x = 0
        while True:
            if len(read_ids) < min_cluster_size * 0.1:
                break
            finish_search = True
            from .device import py_shuffle
            random_candidates = py_shuffle(read_ids).tolist()  # random.shuffle(list(read_ids)), same stream
            found = False
            for s in range(0, len(random_candidates), _PREFETCH):
                block = random_candidates[s:s + _PREFETCH]
                hists = backend.seed_hists(block)  # phase-1 histograms of the next candidates
                for random_point, v in zip(block, _valleys_of_hists(hists)):
                    _, distance_cache, _, _, tail = get_cluster_center(backend, random_point, valley=v)
                    if tail:
                        peel(x, distance_cache, tail)
                        x += 1
                        finish_search = False
                        found = True
                        break
                if found:
                    break
            if finish_search:
                break
    return clusters


def normal(val, mean, std):
    """Sum over features of log(N(val; mean, std) + 1e-7); nan when any std is 0
    (cluster_utils.py:261-268)."""
    with np.errstate(all="ignore"):
        a = np.sqrt(2 * np.pi) * std
        b = np.exp(-0.5 * np.square((val - mean) / std))
        return np.sum(np.log(b / a + 0.0000001))


def _assign_leftovers(profiles, unclassified, cluster_profiles, backend):
    """argmax_k normal(x_r; mean_k, std_k) for every left-over read, first maximum wins,
    nan never wins (cluster_utils.py:309-322).  Returns {read: cluster or None}.  With the
    HIP backend this is one launch of the K5 kernel; other backends (tests) use the
    host formula."""
    keys = list(cluster_profiles.keys())
    if not unclassified or not keys:
        return {r: None for r in unclassified}
    mean = np.stack([cluster_profiles[k]['mean'] for k in keys]).astype(np.float64)
    std = np.stack([cluster_profiles[k]['std'] for k in keys]).astype(np.float64)
    rows = np.fromiter(unclassified, dtype=np.int64, count=len(unclassified))
    out = {}
    if isinstance(backend, HipBackend):
        t = backend.torch
        mean_t = t.from_numpy(mean).to(backend.dev)
        std_t = t.from_numpy(std).to(backend.dev)
        chunk = 1 << 20
        for s in range(0, len(rows), chunk):
            x = t.from_numpy(np.ascontiguousarray(profiles[rows[s:s + chunk]], dtype=np.float64)).to(backend.dev)
            best, _ = backend.ctx.gauss_assign_dev(x, mean_t, std_t)
            best = best.cpu().numpy()
            for r, b in zip(rows[s:s + chunk], best):
                out[int(r)] = None if b < 0 else keys[int(b)]
        return out
    for r in rows:
        max_p, best_c = float('-inf'), None
        for k, m, sd in zip(keys, mean, std):
            p = normal(profiles[r], m, sd)
            if p > max_p:
                max_p, best_c = p, k
        out[int(r)] = best_c
    return out


def perform_binning(output, iterations, min_cluster_size, binreads, reads, backend=None):
    """cluster_utils.py:271-362: clusters -> bins.txt / lengths.txt /
    binning_result.pkl (+ binned_reads/Bin-k.fasta)."""
    latent = _npcache.load(f'{output}/latent.npy')
    logger.info("Clustering algorithm running")
    if backend is None:
        backend = HipBackend()
    clusters = cluster_points(latent, iterations, min_cluster_size, backend=backend)
    clusters_output = {}
    logger.info(f"Detected {len(clusters)} clusters")

    for k, v in clusters.items():
        if len(v) > min_cluster_size:
            clusters_output[len(clusters_output)] = list(map(int, v))
    logger.info(
        f"Detected {len(clusters_output)} clusters with more than {min_cluster_size} points")

    logger.info("Building profiles")
    comp_profiles = _npcache.load(f"{output}/profiles/com_profs.npy")
    cov_profiles = _npcache.load(f"{output}/profiles/cov_profs.npy")
    profiles = np.concatenate([comp_profiles, cov_profiles], axis=1)

    cluster_profiles = {}
    classified_reads = set()
    for k, rs in clusters_output.items():
        vecs = profiles[np.array(rs, dtype=np.int64)]
        classified_reads.update(rs)
        cluster_profiles[k] = {'mean': vecs.mean(axis=0), 'std': vecs.std(axis=0)}

    unclassified_reads = set(range(len(comp_profiles))) - classified_reads
    logger.debug(f"Unclassified points to cluster {len(unclassified_reads)}")
    logger.info("Binning unclassified reads")
    best = _assign_leftovers(profiles, unclassified_reads, cluster_profiles, backend)
    for r in unclassified_reads:
        if best[r] is not None:
            clusters_output[best[r]].append(r)

    logger.info(f"Binning complete with {len(clusters_output)} bins")
    with open(f"{output}/binning_result.pkl", "wb+") as f:
        pickle.dump(clusters_output, f)

    # read -> bin as an array (the reference's dict, built entry by entry, is two million inserts per million reads);
    # a later bin overwrites an earlier one exactly as the dict did
    n_reads = len(comp_profiles)
    bin_of = np.full(n_reads, -1, dtype=np.int64)
    for k, v in clusters_output.items():
        bin_of[np.fromiter(v, dtype=np.int64, count=len(v))] = k

    class _ReadBin:  # read_bin[r]: KeyError for a read no cluster would take, as the reference's dict
        def __getitem__(self, r):
            b = int(bin_of[r]) if 0 <= r < n_reads else -1
            if b < 0:
                raise KeyError(r)
            return b

    read_bin = _ReadBin()

    if binreads:
        if os.path.isdir(f"{output}/binned_reads"):
            shutil.rmtree(f"{output}/binned_reads")
        os.makedirs(f"{output}/binned_reads")
    bin_files = {}

    from . import device as lrb
    if not binreads:
        # only the lengths are needed: no second pass over the sequences
        from . import runners_utils
        lens = runners_utils.read_lengths(reads)
        missing = np.flatnonzero(bin_of[:min(len(lens), n_reads)] < 0)   # KeyError for the first read without a bin
        if missing.size or len(lens) > n_reads:
            raise KeyError(int(missing[0]) if missing.size else n_reads)
        line_of = [f"{b}\n" for b in range(int(bin_of.max()) + 1 if n_reads else 0)]
        with open(f"{output}/bins.txt", "w+") as binout:
            binout.write("".join(map(line_of.__getitem__, bin_of[:len(lens)].tolist())))
        with open(f"{output}/lengths.txt", "w+") as lenout:
            lenout.write("\n".join(map(str, lens.tolist())) + ("\n" if len(lens) else ""))
        return
    r = 0
    with open(f"{output}/bins.txt", "w+") as binout, open(f"{output}/lengths.txt", "w+") as lenout, \
            lrb.FastxReader(reads) as rd:
        for seqs, offs in rd:
            for i in range(len(offs) - 1):
                b = read_bin[r]  # KeyError for a read no cluster would take, as the reference
                binout.write(f"{b}\n")
                lenout.write(f"{int(offs[i + 1] - offs[i])}\n")
                if b not in bin_files:
                    bin_files[b] = open(f"{output}/binned_reads/Bin-{b}.fasta", "w+")
                bin_files[b].write(f">read-{r}\n")
                bin_files[b].write(seqs[int(offs[i]):int(offs[i + 1])].tobytes().decode("latin-1"))
                bin_files[b].write("\n")
                r += 1
    for f in bin_files.values():
        f.close()
