"""BASELINE config 5 at FULL size in the -m gpu suite: `lrbinner.py contigs` on 500 k synthetic contigs
(3.3 GB; ~1.66 M fragments), k = 4 + 15-mer table of the READS + coverage of the fragments + VAE +
HDBSCAN (K6) + the majority vote -- with assertions on every stage's product (scripts/c5_full.py is
the same run as a timing script)."""
import os
import pickle
import subprocess
import sys
from collections import Counter, defaultdict

import numpy as np
import pytest

from helpers import ROOT

pytestmark = pytest.mark.gpu

N_CONTIGS = 500_000


@pytest.fixture(scope="module")
def c5(tmp_path_factory):
    scratch = "/dev/shm" if os.path.isdir("/dev/shm") else None
    import tempfile
    tmp = tempfile.mkdtemp(dir=scratch, prefix="lrb_c5_")
    rng = np.random.default_rng(5)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    n_genomes, glen = 10, 4_000_000
    genomes = []
    for g in range(n_genomes):
        p = rng.dirichlet(np.full(4, 6.0))
        genomes.append(letters[(rng.random(glen)[:, None] > np.cumsum(p)[None, :]).sum(1).clip(0, 3)])
    cov = np.array([5, 7, 9, 12, 16, 21, 28, 37, 48, 60], dtype=np.float64)
    origin = rng.choice(n_genomes, size=N_CONTIGS, p=cov / cov.sum())
    lens = np.clip(rng.lognormal(8.6, 0.6, N_CONTIGS).astype(np.int64), 1500, 60000)
    starts = (rng.random(N_CONTIGS) * (glen - lens)).astype(np.int64)
    contigs, reads = os.path.join(tmp, "contigs.fasta"), os.path.join(tmp, "reads.fasta")
    with open(contigs, "wb", buffering=1 << 24) as f:
        for i in range(N_CONTIGS):
            f.write(b">contig_%d\n" % i)
            f.write(genomes[origin[i]][starts[i]:starts[i] + lens[i]].tobytes())
            f.write(b"\n")
    n_reads, L = 200_000, 8000
    rorigin = rng.choice(n_genomes, size=n_reads, p=cov / cov.sum())
    rstarts = rng.integers(0, glen - L, size=n_reads)
    with open(reads, "wb", buffering=1 << 24) as f:
        for i in range(n_reads):
            f.write(b">r%d\n" % i)
            f.write(genomes[rorigin[i]][rstarts[i]:rstarts[i] + L].tobytes())
            f.write(b"\n")
    out = os.path.join(tmp, "out")
    cmd = [sys.executable, os.path.join(ROOT, "lrbinner.py"), "contigs", "-r", reads, "-c", contigs, "-o", out,
           "-k", "4", "--ae-dims", "8", "--ae-epochs", "50", "--cuda", "-t", "32"]
    subprocess.run(cmd, check=True, cwd=ROOT, env=dict(os.environ, LRB_SEED="5"))
    yield {"out": out, "genomes": genomes, "origin": origin, "lens": lens, "starts": starts}
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)


def _fragments_of(length):
    """split_contigs (runners_utils.py:53-75): >= 5000 bp -> windows of 2500 plus the last 2500 bp."""
    if length >= 5000:
        return [(x, min(x + 2500, length)) for x in range(0, length, 2500)] + [(length - 2500, length)]
    return [(0, length)]


def test_c5_fragments_and_profiles(c5):
    """Fragment count and parentage from the split rule; the k=4 rows of sampled fragments equal the
    oracle's counts / total printed with %f (the npy holds the six-decimal values, pipelines.py:315-321)."""
    from oracle import oracle as orc
    out, lens = c5["out"], c5["lens"]
    per = np.where(lens >= 5000, -(-lens // 2500) + 1, 1)
    n_frag = int(per.sum())
    com = np.load(os.path.join(out, "profiles/com_profs.npy"), mmap_mode="r")
    cov = np.load(os.path.join(out, "profiles/cov_profs.npy"), mmap_mode="r")
    lat = np.load(os.path.join(out, "latent.npy"))
    assert com.shape == (n_frag, 136) and cov.shape == (n_frag, 32) and lat.shape == (n_frag, 8) and lat.dtype == np.float32
    assert np.isfinite(lat).all()
    parent = pickle.load(open(os.path.join(out, "profiles/fragment_parent.pkl"), "rb"))
    first = np.concatenate([[0], np.cumsum(per)[:-1]])
    rng = np.random.default_rng(1)
    pick = np.r_[0, N_CONTIGS - 1, rng.choice(N_CONTIGS, 62, replace=False)]
    seqs, rows = [], []
    for c in pick:
        assert parent[int(first[c])] == f"contig_{c}" and parent[int(first[c] + per[c] - 1)] == f"contig_{c}"
        g = c5["genomes"][c5["origin"][c]][c5["starts"][c]:c5["starts"][c] + lens[c]]
        for j, (a, b) in enumerate(_fragments_of(int(lens[c]))):
            seqs.append(g[a:b].tobytes())
            rows.append(int(first[c]) + j)
    buf, offs = orc.concat(seqs)
    counts, totals = orc.count_kmers(buf, offs, 4)
    want = np.array([[float("%f" % (c / t)) for c in row] for row, t in zip(counts, totals.astype(np.float64))])
    assert np.array_equal(np.asarray(com[rows]), want)
    assert np.allclose(np.asarray(cov[rows]).sum(1), 1.0, atol=1e-4 * 32)
    # coverage rows: the pipeline tallies the 1.66 M fragments as sweeps over ranges of resident batches laid end
    # to end (lrb_packed_cov_hist_many); here the sampled fragments go through the OTHER K3 path -- gathers from a
    # table rebuilt from the reads file -- and have to print the same six decimals
    from lrbinner_amd import device as lrb, runners_utils as ru
    from helpers import parse_profile_text
    ctx = lrb.Context(0)
    table = ctx.alloc_table()
    try:
        for seqs_b, offs_b in ru._batches(os.path.join(os.path.dirname(out), "reads.fasta"), 8):
            ctx.k15_accumulate(seqs_b, offs_b, table)
        ctx.k15_mirror(table)
        hist, sums = ctx.cov_hist(buf, offs, table, 10, 32)
    finally:
        ctx.free(table)
    assert (sums > 0).all()
    assert np.array_equal(np.asarray(cov[rows]), parse_profile_text(lrb.format_cov(hist, sums, threads=2)))


def test_c5_vote_follows_the_reference_walk(c5):
    """bins.txt against an independent restatement of cluster_utils.py:496-520 on the labels HDBSCAN
    gives for the run's own latent.npy (K6 is deterministic): same contigs, same bins, same order."""
    from lrbinner_amd import device as lrb
    out = c5["out"]
    lat = np.load(os.path.join(out, "latent.npy"))
    labels = lrb.Context(0).hdbscan(lat, min_cluster_size=250)
    parent = pickle.load(open(os.path.join(out, "profiles/fragment_parent.pkl"), "rb"))
    clusters = defaultdict(list)
    for i, c in enumerate(labels.tolist()):
        if c != -1:
            clusters[c].append(i)
    assert len(clusters) >= 5
    parent_clusters = defaultdict(list)
    for c, idx in clusters.items():
        for i in idx:
            parent_clusters[parent[i]].append(c)
    want = [(contig, Counter(v).most_common()[0][0]) for contig, v in parent_clusters.items()]
    got = [tuple(l.split("\t")) for l in open(os.path.join(out, "bins.txt")).read().splitlines()]
    assert [(a, str(b)) for a, b in want] == got
    # the bins mean something: contigs of one genome share a bin (purity against the genome of origin)
    by_bin = defaultdict(list)
    for cid, b in got:
        by_bin[b].append(c5["origin"][int(cid.split("_")[1])])
    pure = sum(np.bincount(v).max() for v in by_bin.values())
    # (the VAE's float atomics make runs repeat only statistically: 0.96-0.98 as a rule, one run in ten lower; a
    # binning that had lost the signal would sit near the largest genome's share, 0.25)
    assert len(got) > 0.5 * N_CONTIGS and pure / len(got) > 0.90, (len(got), pure / len(got), len(by_bin))


def _label_delta(a, b):
    """(points labelled differently after matching b's clusters to a's by largest overlap, clusters of a, clusters of b,
    noise points of a, noise points of b)"""
    ca, cb = sorted(set(a.tolist()) - {-1}), sorted(set(b.tolist()) - {-1})
    remap = np.full(max(cb, default=-1) + 2, -2, np.int64)
    for c in cb:
        inside = a[b == c]
        inside = inside[inside >= 0]
        remap[c] = np.bincount(inside).argmax() if len(inside) else -2
    bm = np.where(b >= 0, remap[np.maximum(b, 0)], -1)
    return int((bm != a).sum()), len(ca), len(cb), int((a == -1).sum()), int((b == -1).sum())


def test_c5_core_distance_convention_delta(c5):
    """How far apart the two core-distance conventions are on the run's own 1.66 M latents: the
    min_samples-th neighbour WITH the point itself (sklearn, the hdbscan package's Prim's paths) against the
    min_samples-th OTHER point (the package's Boruvka paths = the reference's call; the library default).  At k = 250
    the two differ at cluster borders only: points, clusters and -- after the majority vote -- contigs are counted,
    written to gpurun_out/r04_hdb_convention_delta.json, and bounded."""
    import json
    from lrbinner_amd import device as lrb
    from lrbinner_amd.pipelines import contig_votes
    out = c5["out"]
    lat = np.load(os.path.join(out, "latent.npy"))
    parent = pickle.load(open(os.path.join(out, "profiles/fragment_parent.pkl"), "rb"))
    ctx = lrb.Context(0)
    incl = ctx.hdbscan(lat, min_cluster_size=250, core_excludes_self=False)
    excl = ctx.hdbscan(lat, min_cluster_size=250, core_excludes_self=True)
    assert np.array_equal(excl, ctx.hdbscan(lat, min_cluster_size=250))          # the default
    diff, ca, cb, na, nb = _label_delta(incl, excl)
    va, vb = contig_votes(incl, parent), contig_votes(excl, parent)
    # contig bins compared through the fragment-level cluster matching: a contig "moves" when it is binned under one
    # convention only, or lands in clusters that do not correspond
    both = set(va) & set(vb)
    pairs = Counter((va[c], vb[c]) for c in both)
    best = {}
    for (x, y), k in pairs.items():
        if k > best.get(y, (None, 0))[1]:
            best[y] = (x, k)
    moved = sum(1 for c in both if best[vb[c]][0] != va[c]) + len(set(va) ^ set(vb))
    rec = {"fragments": int(len(lat)), "points_labelled_differently": diff, "clusters_incl_self": ca,
           "clusters_excl_self": cb, "noise_incl_self": na, "noise_excl_self": nb, "contigs_binned_incl_self": len(va),
           "contigs_binned_excl_self": len(vb), "contigs_whose_bin_differs": moved}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rec, open(os.path.join(ROOT, "gpurun_out", "r04_hdb_convention_delta_c5.json"), "w"), indent=1)
    print(rec)
    assert abs(ca - cb) <= max(2, ca // 10) and diff <= 0.02 * len(lat) and moved <= 0.02 * max(len(va), 1), rec


def test_c5_pruned_hdbscan_equals_brute_force_on_the_run_latents(c5, monkeypatch):
    """On 200 k of the run's own fragment latents the spatially pruned kernels give the brute-force
    core distances and spanning tree bit for bit."""
    import torch
    from lrbinner_amd import device as lrb
    lat = np.load(os.path.join(c5["out"], "latent.npy"))
    sub = np.ascontiguousarray(lat[:: max(1, len(lat) // 200_000)][:200_000])
    assert len(sub) == 200_000
    ctx = lrb.Context(0, use_torch_stream=True)
    xt = torch.from_numpy(sub).cuda()
    monkeypatch.setenv("LRB_HDB_BRUTE", "1")
    core0 = ctx.hdb_core_dist_dev(xt, 250)
    u0, v0, w0, r0 = ctx.hdb_mst_dev(xt, core0)
    monkeypatch.setenv("LRB_HDB_BRUTE", "0")
    core1 = ctx.hdb_core_dist_dev(xt, 250)
    u1, v1, w1, r1 = ctx.hdb_mst_dev(xt, core1)
    assert torch.equal(core0.view(torch.int32), core1.view(torch.int32))
    assert np.array_equal(u0, u1) and np.array_equal(v0, v1) and np.array_equal(w0.view(np.uint32), w1.view(np.uint32))


def test_c5_labels_agree_with_sklearn_hdbscan_on_a_subset(c5):
    """The reference's `hdbscan` package is not in the image (parity unpinned for this row); the offline
    implementation of the same published algorithm is sklearn.cluster.HDBSCAN.  On 40 k of the run's own
    fragment latents K6's labels agree with it: same number of clusters, adjusted Rand index >= 0.99 over the
    points both call clustered."""
    sk = pytest.importorskip("sklearn.cluster")
    from helpers import adjusted_rand
    from lrbinner_amd import device as lrb
    lat = np.load(os.path.join(c5["out"], "latent.npy"))
    sub = np.ascontiguousarray(lat[:: max(1, len(lat) // 40_000)][:40_000])
    ours = lrb.Context(0).hdbscan(sub, min_cluster_size=250, core_excludes_self=False)   # sklearn's convention
    ref = sk.HDBSCAN(min_cluster_size=250, algorithm="brute", copy=True).fit_predict(sub.astype(np.float64))
    assert len(set(ours.tolist()) - {-1}) == len(set(ref.tolist()) - {-1})
    both = (ours >= 0) & (ref >= 0)
    assert both.mean() > 0.5 and abs((ours >= 0).mean() - (ref >= 0).mean()) < 0.02
    assert adjusted_rand(ours[both], ref[both]) >= 0.99
