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


# ---------------------------------------------------------------------------
# deterministic synthetic metagenome (stand-in for the Sim-8 set, which is an
# external download): genomes with distinct order-2 Markov composition, spread
# abundances, noisy long reads, ground-truth labels.
# ---------------------------------------------------------------------------
def synth_metagenome(seed=2024, n_genomes=4, genome_len=300_000, read_len=3000,
                     coverages=(8.0, 16.0, 32.0, 64.0), err=0.12, window=5000,
                     spread_frac=0.1, spread_range=(1.5, 160.0), dirichlet=1.5):
    """-> (list of read bytes, labels int array).  Same output for the same arguments
    on every platform (numpy Generator streams are stable).

    Every genome has its own abundance (the coverage signal) and its own order-2 Markov
    composition.  A tenth of each genome's windows is sampled at a log-uniformly spread
    local coverage instead, so every cluster owns a few reads in every coverage bin:
    without that the reference's left-over assignment is nan for every cluster and it
    dies with KeyError."""
    rng = np.random.default_rng(seed)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    reads, labels = [], []
    for g in range(n_genomes):
        # order-2 Markov chain with a genome-specific transition table
        trans = rng.dirichlet(np.ones(4) * dirichlet, size=16)
        cum = np.cumsum(trans, axis=1)
        u = rng.random(genome_len)
        seq = np.zeros(genome_len, dtype=np.int64)
        seq[:2] = rng.integers(0, 4, 2)
        for i in range(2, genome_len):
            seq[i] = np.searchsorted(cum[seq[i - 2] * 4 + seq[i - 1]], u[i])
        seq = np.minimum(seq, 3)
        nwin = genome_len // window
        local = np.full(nwin, float(coverages[g % len(coverages)]))
        spread = rng.random(nwin) < spread_frac
        lo, hi = np.log(spread_range[0]), np.log(spread_range[1])
        local[spread] = np.exp(rng.uniform(lo, hi, int(spread.sum())))
        starts = []
        for w in range(nwin):
            k = rng.poisson(local[w] * window / read_len)
            starts.append(rng.integers(w * window, (w + 1) * window, k))
        starts = np.concatenate(starts)
        starts = np.minimum(starts, genome_len - read_len)
        for s in starts:
            r = seq[s:s + read_len].copy()
            m = rng.random(read_len) < err          # substitutions
            r[m] = rng.integers(0, 4, int(m.sum()))
            if rng.random() < 0.5:                  # random strand
                r = (3 - r)[::-1]                   # A<->T, C<->G with ACGT = 0123
            reads.append(alpha[r].tobytes())
            labels.append(g)
    order = rng.permutation(len(reads))
    return [reads[i] for i in order], np.array(labels)[order]


SIM8_LENS_MBP = (5.0, 4.0, 3.5, 3.0, 2.5, 2.0, 1.5, 1.0)
SIM8_COVS = (5.0, 8.0, 12.0, 17.0, 24.0, 33.0, 45.0, 60.0)
# GC content of the eight genomes, 3.5 % apart.  At 2 % steps both the reference and this build merge two
# neighbours in about one run in three (7 bins, F1 96) -- same behaviour, but a gate on five runs would then
# hang on a coin; at this spacing every run of either finds the eight (scripts/sim8_explore.py).
SIM8_GC = (0.36, 0.395, 0.43, 0.465, 0.50, 0.535, 0.57, 0.61)


def synth_sim8(seed=8, scale=1.0, read_len=10_000, p_sub=0.04, p_del=0.03, p_ins=0.03, conc=300.0):
    """The accuracy stand-in SURVEY.md 8(d) describes for the Sim-8 set (an external download):
    eight genomes of 1-5 Mbp, each an order-3 Markov chain around its own GC content, abundances
    5x-60x, 10 kb reads from a random strand with ~10 % substitution/indel noise, ground truth kept.
    -> (list of read bytes, labels int array); ~40 k reads at scale 1.  Deterministic: numpy
    Generator streams for the tables and the read starts, splitmix64 (oracle/lrb_oracle.c) for
    the genomes and the noise.  `scale` multiplies the genome lengths (tests use < 1)."""
    import sys
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle import oracle as orc
    rng = np.random.default_rng(seed)
    reads, labels = [], []
    for g, (mbp, cov, gc) in enumerate(zip(SIM8_LENS_MBP, SIM8_COVS, SIM8_GC)):
        glen = int(mbp * 1e6 * scale)
        base = np.array([(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2])        # A C G T
        trans = rng.dirichlet(base * conc, size=64)
        genome = orc.synth_markov(seed * 1000 + g, 3, np.cumsum(trans, axis=1), glen)
        n = int(round(glen * cov / read_len))
        starts = rng.integers(0, glen - read_len, size=n)
        strand = rng.random(n) < 0.5
        for i in range(n):
            s = int(starts[i])
            reads.append(orc.synth_read((seed << 40) + (g << 32) + i, genome[s:s + read_len],
                                        p_sub, p_del, p_ins, bool(strand[i])))
            labels.append(g)
    order = rng.permutation(len(reads))
    return [reads[i] for i in order], np.array(labels)[order]


# C1 of BASELINE.json on its own flags and at its own size (README.md:73: -k 3 -bc 10 -bs 32 --ae-dims 4 -mbs 5000 on
# the 432,333 reads of Sim-8): eight genomes of 100-600 kbp at 550x-3,100x, so that the 15-mer counts of a 10 %-noise
# read (0.9^15 = 21 % of the coverage) run into the hundreds and a histogram of 10 bins of width 32 says something
C1_LENS_KBP = (100, 150, 200, 280, 360, 440, 520, 600)
C1_COVS = (3100.0, 2400.0, 1900.0, 1500.0, 1200.0, 950.0, 740.0, 550.0)
C1_READS = 432_333


def synth_sim8_c1(seed=8, n_reads=C1_READS, read_len=10_000, p_sub=0.04, p_del=0.03, p_ins=0.03, conc=300.0):
    """The Sim-8 stand-in at the README's own size and coverage: EXACTLY n_reads reads of 10 kb from eight order-3
    Markov genomes (GC contents of synth_sim8) of 100-600 kbp; the coverages above are rescaled together so that the
    read counts add up to n_reads.  Same generators as synth_sim8, so the same bytes everywhere.
    -> (list of read bytes, labels int array)"""
    import sys
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle import oracle as orc
    rng = np.random.default_rng(seed)
    glens = [int(k * 1000) for k in C1_LENS_KBP]
    want = np.array([g * c / read_len for g, c in zip(glens, C1_COVS)])
    counts = np.floor(want * (n_reads / want.sum())).astype(np.int64)
    counts[-1] += n_reads - int(counts.sum())
    reads, labels = [], []
    for g, (glen, n, gc) in enumerate(zip(glens, counts.tolist(), SIM8_GC)):
        base = np.array([(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2])        # A C G T
        trans = rng.dirichlet(base * conc, size=64)
        genome = orc.synth_markov(seed * 1000 + 100 + g, 3, np.cumsum(trans, axis=1), glen)
        starts = rng.integers(0, glen - read_len, size=n)
        strand = rng.random(n) < 0.5
        for i in range(n):
            s = int(starts[i])
            reads.append(orc.synth_read((seed << 40) + ((g + 16) << 32) + i, genome[s:s + read_len],
                                        p_sub, p_del, p_ins, bool(strand[i])))
            labels.append(g)
    order = rng.permutation(len(reads))
    return [reads[i] for i in order], np.array(labels)[order]


# An accuracy set at C1's size on which the method STRAINS (round 4): the eighth genome is a STRAIN of the seventh --
# the same sequence with C1H_STRAIN_DIV point substitutions -- at three times its abundance (300x / 900x, inside the
# range of the README's histogram: 10 bins of width 32).  Composition cannot tell the two apart; the 15-mer coverage
# histogram has to (scripts/c1_hard_explore.py: 8 bins F1 99.87, or the pair merged: 7 bins F1 97.1; the reference: 8 bins
# three times of three, this build's CLI: ten times of ten).  Variants tried there: GC contents 2 % apart in pairs as well (a second, unrelated merge
# in a third of the runs: F1 94), 3-6 % divergence (the pair merged in most or all runs), 200x / 1000x (always merged).
C1H_GC = (0.36, 0.395, 0.43, 0.465, 0.50, 0.535, 0.57, 0.57)
C1H_LENS_KBP = (100, 150, 200, 280, 360, 440, 520, 520)
C1H_COVS = (3100.0, 2400.0, 1900.0, 1500.0, 1200.0, 950.0, 300.0, 900.0)
C1H_STRAIN_OF = {7: 6}
C1H_STRAIN_DIV = 0.10


def synth_sim8_c1_hard(seed=8, n_reads=C1_READS, read_len=10_000, p_sub=0.04, p_del=0.03, p_ins=0.03, conc=300.0,
                       gcs=C1H_GC, lens_kbp=C1H_LENS_KBP, covs=C1H_COVS, strain_of=C1H_STRAIN_OF, strain_div=C1H_STRAIN_DIV):
    """synth_sim8_c1 with close GC pairs and strains (genome g = genome strain_of[g] with strain_div substitutions).
    -> (list of read bytes, labels int array)"""
    import sys
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle import oracle as orc
    rng = np.random.default_rng(seed)
    glens = [int(k * 1000) for k in lens_kbp]
    want = np.array([g * c / read_len for g, c in zip(glens, covs)])
    counts = np.floor(want * (n_reads / want.sum())).astype(np.int64)
    counts[-1] += n_reads - int(counts.sum())
    genomes, reads, labels = [], [], []
    for g, (glen, n, gc) in enumerate(zip(glens, counts.tolist(), gcs)):
        base = np.array([(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2])        # A C G T
        trans = rng.dirichlet(base * conc, size=64)
        if g in strain_of:
            letters = np.frombuffer(b"ACGT", dtype=np.uint8)      # (in ascending byte order)
            genome = np.array(genomes[strain_of[g]][:glen], dtype=np.uint8, copy=True)
            hit = rng.random(glen) < strain_div
            # a substituted base is one of the three OTHER letters
            genome[hit] = letters[(np.searchsorted(letters, genome[hit]) + rng.integers(1, 4, int(hit.sum()))) % 4]
        else:
            genome = orc.synth_markov(seed * 1000 + 300 + g, 3, np.cumsum(trans, axis=1), glen)
        genomes.append(genome)
        starts = rng.integers(0, glen - read_len, size=n)
        strand = rng.random(n) < 0.5
        for i in range(n):
            s0 = int(starts[i])
            reads.append(orc.synth_read((seed << 40) + ((g + 32) << 32) + i, genome[s0:s0 + read_len],
                                        p_sub, p_del, p_ins, bool(strand[i])))
            labels.append(g)
    order = rng.permutation(len(reads))
    return [reads[i] for i in order], np.array(labels)[order]


def synth_block_mixture(n_reads, read_len=5000, seed=8, n_genomes=8, glen=1_500_000, err=0.10):
    """The data set of profiles/r01_e2e_pipeline.json (scripts/e2e_pipeline_scale.py in round 1): every
    genome is a patchwork of 5 kb blocks drawn from TWO order-0 base compositions, reads are 5 kb
    windows with 10 % substitutions.  Kept to show what that run measured: a 5 kb read lies mostly in
    one or two blocks, so each genome presents up to three composition modes (p, q, mixtures) and a
    composition-driven binner legitimately splits it -- many bins, high precision, low recall -- the
    reference included (tests/golden/e2e_reference_blocks.json).  -> (reads, labels)"""
    rng = np.random.default_rng(seed)
    cov = np.array([4, 6, 9, 13, 19, 28, 41, 60], dtype=np.float64)[:n_genomes]
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    genomes = []
    for g in range(n_genomes):
        p, q = rng.dirichlet(np.full(4, 6.0)), rng.dirichlet(np.full(4, 6.0))
        blocks = rng.random(glen // 5000 + 1) < 0.5
        prob = np.where(np.repeat(blocks, 5000)[:glen, None], p[None, :], q[None, :])
        u = rng.random(glen)
        genomes.append(letters[(u[:, None] > np.cumsum(prob, axis=1)).sum(1).clip(0, 3)])
    origin = rng.choice(n_genomes, size=n_reads, p=cov / cov.sum())
    starts = rng.integers(0, glen - read_len, size=n_reads)
    reads = []
    for i in range(n_reads):
        r = genomes[origin[i]][starts[i]:starts[i] + read_len].copy()
        sub = rng.random(read_len) < err
        r[sub] = letters[rng.integers(0, 4, size=int(sub.sum()))]
        reads.append(r.tobytes())
    return reads, origin


def write_fasta(path, reads):
    with open(path, "wb") as f:
        for i, r in enumerate(reads):
            f.write(b">read%d\n" % i + r + b"\n")


def binning_scores(bins, truth):
    """Precision / recall / F1 (percent) as eval.py:37-45 computes them from the
    bins x species count matrix."""
    bins, truth = np.asarray(bins), np.asarray(truth)
    bi = {b: i for i, b in enumerate(sorted(set(bins.tolist())))}
    ti = {t: i for i, t in enumerate(sorted(set(truth.tolist())))}
    m = np.zeros((len(bi), len(ti)), dtype=np.int64)
    for b, t in zip(bins.tolist(), truth.tolist()):
        m[bi[b], ti[t]] += 1
    total = m.sum()
    precision = m.max(axis=1).sum() / total * 100
    recall = m.max(axis=0).sum() / total * 100
    f1 = 2 * precision * recall / (precision + recall)
    return float(precision), float(recall), float(f1), len(bi)


def merged_genomes(bins, truth):
    """The groups of genomes that ended in ONE bin: a genome's home is the bin most of its reads sit in; genomes that share
    a home are merged.  -> sorted list of lists, e.g. [[6, 7]] (the strain pair of helpers.synth_sim8_c1_hard) or []."""
    bins, truth = np.asarray(bins), np.asarray(truth)
    groups = {}
    for g in sorted(set(truth.tolist())):
        vals, cnt = np.unique(bins[truth == g], return_counts=True)
        groups.setdefault(int(vals[np.argmax(cnt)]), []).append(int(g))
    return sorted(v for v in groups.values() if len(v) > 1)


def np_planes(buf, offs):
    """Bit-plane form (include/lrb_hip.h, lrb_kmer_counts3_dev): uint32[2*mask_words],
    {H, L} per 32-base block at words 2*(mask_off[r]+b), +1; first base in bit 31."""
    n = len(offs) - 1
    lens = np.diff(offs).astype(np.int64)
    mw = ((-(-lens // 32) + 3) // 4) * 4 + 4
    mo = np.zeros(n + 1, np.int64)
    mo[1:] = np.cumsum(mw)
    planes = np.zeros(2 * int(mo[-1]), np.uint32)
    for r in range(n):
        s = buf[int(offs[r]):int(offs[r + 1])].astype(np.uint32)
        if len(s) == 0:
            continue
        code = (s >> 1) & 3
        i = np.arange(len(s))
        sh = (31 - (i % 32)).astype(np.uint32)
        np.bitwise_or.at(planes, 2 * (int(mo[r]) + i // 32), ((code >> 1) << sh).astype(np.uint32))
        np.bitwise_or.at(planes, 2 * (int(mo[r]) + i // 32) + 1, ((code & 1) << sh).astype(np.uint32))
    return planes


def adjusted_rand(a, b):
    """Adjusted Rand index of two labelings (Hubert & Arabie 1985), noise (-1) as a class."""
    a = np.asarray(a)
    b = np.asarray(b)
    _, ai = np.unique(a, return_inverse=True)
    _, bi = np.unique(b, return_inverse=True)
    cont = np.zeros((ai.max() + 1, bi.max() + 1), dtype=np.int64)
    np.add.at(cont, (ai, bi), 1)
    comb = lambda x: x * (x - 1) / 2.0
    s_ij = comb(cont).sum()
    s_a = comb(cont.sum(1)).sum()
    s_b = comb(cont.sum(0)).sum()
    total = comb(len(a))
    expected = s_a * s_b / total
    mx = 0.5 * (s_a + s_b)
    return 1.0 if mx == expected else (s_ij - expected) / (mx - expected)


# ---- statistics of the hard accuracy set (tests/test_gpu_sim8.py, tests/test_host_logic.py) ----
import json  # noqa: E402


def binom_pmf(k, n, p):
    from math import comb
    return comb(n, k) * p ** k * (1 - p) ** (n - k)


def binom_sf(k, n, p):
    """P(X > k), X ~ Binomial(n, p)"""
    return sum(binom_pmf(i, n, p) for i in range(k + 1, n + 1))


def cp_upper(k, n, conf=0.95):
    """one-sided Clopper-Pearson upper bound of a rate seen k times in n"""
    if k >= n:
        return 1.0
    lo, hi = k / n, 1.0
    for _ in range(60):
        mid = (lo + hi) / 2
        if 1 - binom_sf(k, n, mid) > 1 - conf:     # P(X <= k | mid) still above the tail: the bound lies further up
            lo = mid
        else:
            hi = mid
    return hi


def fisher_one_sided(k1, n1, k2, n2):
    """P(at least k1 of the k1 + k2 events fall in sample 1 | the two samples share one rate): hypergeometric tail"""
    from math import comb
    K, N = k1 + k2, n1 + n2
    return sum(comb(n1, i) * comb(n2, K - i) for i in range(k1, min(K, n1) + 1) if K - i <= n2) / comb(N, K)


def _outcome_class(run):
    """'none' / 'strain' ([6, 7] share a bin: F1 97.1-97.2) / 'gc' ([5, 7]: F1 92.3-92.4) / 'both' (6 bins, F1 89.2) of a
    whole run on helpers.synth_sim8_c1_hard -- from the recorded merged groups, or (runs recorded before round 6) from the
    number of bins and the F1 the merged pair costs."""
    if "merged" in run:
        m = [sorted(g) for g in run["merged"]]
        if not m:
            return "none"
        flat = sorted(x for g in m for x in g)
        if flat == [6, 7]:
            return "strain"
        if flat == [5, 7]:
            return "gc"
        return "both" if set(flat) >= {5, 6, 7} else "other"
    if run["bins"] >= 8:
        return "none"
    if run["bins"] == 7:
        return "strain" if run["f1"] > 95.0 else "gc"
    return "both"


def hard_set_statistics():
    """The measured outcome distributions on helpers.synth_sim8_c1_hard and what follows from them.
    reference: tests/golden/e2e_reference_c1_hard.json (the REFERENCE's pipeline, build container, one run per seed; from
               round 6 with the merged groups and helpers.latent_pair_stats of every run's latent.npy);
    this build: profiles/r06_c1hard_runs_100.json (100 seeded whole runs on the MI355X, library defaults: outcome, pair
               statistics, the same latents clustered again under three more search seeds) and the 120 runs of round 5
               (profiles/r05_c1_hard_rates.json: outcome only)."""
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    ref = json.load(open(os.path.join(gold, "e2e_reference_c1_hard.json")))
    ours = json.load(open(os.path.join(ROOT, "profiles", "r06_c1hard_runs_100.json")))
    old = json.load(open(os.path.join(ROOT, "profiles", "r05_c1_hard_rates.json")))
    ref_runs, our_runs = ref["runs"], ours["runs"]
    k_ref, n_ref = sum(r["bins"] < 8 for r in ref_runs), len(ref_runs)
    k_b, n_b = sum(r["bins"] < 8 for r in our_runs), len(our_runs)
    classes = {}
    for cls in ("strain", "gc", "both"):
        kr = sum(_outcome_class(r) == cls for r in ref_runs)
        kb = sum(_outcome_class(r) == cls for r in our_runs)
        kb_all = kb + sum(_outcome_class(r) == cls for r in old["default_mode"])
        classes[cls] = {"ref": kr, "n_ref": n_ref, "build": kb, "n_build": n_b, "fisher_p_build_worse": fisher_one_sided(kb, n_b, kr, n_ref),
                        "build_r5_and_r6": kb_all, "n_build_r5_and_r6": n_b + len(old["default_mode"]),
                        "fisher_p_build_worse_r5_and_r6": fisher_one_sided(kb_all, n_b + len(old["default_mode"]), kr, n_ref)}
    return {"ref_runs": ref_runs, "our_runs": our_runs, "k_ref": k_ref, "n_ref": n_ref, "k_b": k_b, "n_b": n_b, "classes": classes,
            "rate_ref_upper95": cp_upper(k_ref, n_ref), "rate_build_upper95": cp_upper(k_b, n_b),
            "fisher_p_build_worse": fisher_one_sided(k_b, n_b, k_ref, n_ref),
            "mean_f1_ref": float(np.mean([r["f1"] for r in ref_runs])), "mean_f1_build": float(np.mean([r["f1"] for r in our_runs]))}


# ---- a continuous separation statistic on latent.npy (round 6: what 2-of-25 events cannot resolve) ----
C1H_PAIRS = {"strain": (6, 7), "gc01": (0, 1), "gc12": (1, 2), "gc23": (2, 3), "gc34": (3, 4), "gc45": (4, 5), "gc56": (5, 6),
             "gc57": (5, 7)}
C1_PAIRS = {f"gc{i}{i + 1}": (i, i + 1) for i in range(7)}      # helpers.synth_sim8_c1: neighbours in GC content


def latent_pair_stats(latent, labels, pairs=None):
    """Per genome pair (a, b) of a labelled latent.npy, in the geometry the cluster search works in (rows scaled to unit
    length, cluster_utils.py:31-42 -- its distance is (1 - cos) / 2):
      dprime  angle between the two centroids / pooled RMS angle of the members to their own centroid (a d' statistic:
              how many within-genome spreads apart the two genomes lie);
      valley  the ratio find_valley_ratio (cluster_utils.py:87-133; the oracle's restatement) returns for the distance
              histogram over the READS OF THE TWO GENOMES from the medoid of a -- density at the valley between the two
              genomes / density just past a's peak; 1.0 when the scan finds no valley (the pair reads as one cluster).
    -> {pair name: {"dprime": float, "valley": float, "valley_ba": float}}"""
    import sys
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle import np_cluster as nc
    pairs = C1H_PAIRS if pairs is None else pairs
    lat = np.asarray(latent, dtype=np.float64)
    labels = np.asarray(labels)
    nrm = np.linalg.norm(lat, axis=1)
    nrm[nrm == 0] = 1.0
    u = lat / nrm[:, None]
    cen, spread, medoid = {}, {}, {}
    for g in sorted(set(int(x) for pr in pairs.values() for x in pr)):
        idx = np.flatnonzero(labels == g)
        c = u[idx].mean(axis=0)
        c /= np.linalg.norm(c)
        cosv = np.clip(u[idx] @ c, -1.0, 1.0)
        cen[g] = c
        spread[g] = float(np.sqrt(np.mean(np.arccos(cosv) ** 2)))
        medoid[g] = int(idx[int(np.argmax(cosv))])
    m32 = nc.normalize(np.asarray(latent, dtype=np.float32))

    def valley(a, b):
        idx = np.flatnonzero((labels == a) | (labels == b))
        sub = m32[idx]
        seed = int(np.searchsorted(idx, medoid[a]))
        _, prof = nc._seed_profile(sub, seed)
        return 1.0 if prof[0] is False else float(prof[0])

    out = {}
    for name, (a, b) in pairs.items():
        theta = float(np.arccos(np.clip(cen[a] @ cen[b], -1.0, 1.0)))
        pooled = float(np.sqrt(0.5 * (spread[a] ** 2 + spread[b] ** 2)))
        out[name] = {"dprime": theta / pooled if pooled > 0 else float("inf"), "valley": valley(a, b), "valley_ba": valley(b, a)}
    return out
