"""Host logic that never touches the GPU: Checkpointer transitions and split_contigs vs
vectors produced by the reference's own runners_utils (tests/golden/py_host.json),
reader fuzzing against the oracle reader, CLI parsing."""
import hashlib
import json
import os
import re

import numpy as np
import pytest

from helpers import golden_path
from oracle import oracle as orc
from lrbinner_amd import device
from lrbinner_amd import runners_utils as ru


@pytest.fixture(scope="module")
def gh():
    return json.load(open(golden_path("py_host.json")))


def test_checkpointer_transitions_match_reference(gh, tmp_path):
    cp = ru.Checkpointer(str(tmp_path / "ck"))
    for op, stage, params, expected in gh["checkpoint_trace"]:
        if op == "run?":
            assert cp.should_run_step(stage, params) == expected, (stage, params)
        else:
            cp.log(stage, params)
            assert sorted(cp.completed) == expected, (stage, params)
    assert sorted(ru.Checkpointer(str(tmp_path / "ck"), True).completed) == gh["checkpoint_reload"]
    assert ru.Checkpointer(str(tmp_path / "ck")).completed == {}       # not resuming: start empty


def test_split_contigs_matches_reference(gh, tmp_path):
    os.makedirs(tmp_path / "fragments")
    fa = tmp_path / "contigs.fasta"
    fa.write_text(gh["contigs_fasta"])
    groups, parent = ru.split_contigs(str(fa), str(tmp_path))
    assert dict(groups) == gh["groups"]
    assert {str(k): v for k, v in parent.items()} == gh["parent"]
    sha = hashlib.sha256((tmp_path / "fragments" / "contigs.fasta").read_bytes()).hexdigest()
    assert sha == gh["fragments_fasta_sha"]


def _random_fastx(rng):
    """A syntactically loose FASTA/FASTQ blob: CRLF, blank lines, lower case, stray
    header characters inside lines, missing final newline, truncated FASTQ tails."""
    out = bytearray()
    alpha = np.frombuffer(b"ACGTNacgt", dtype=np.uint8)
    fastq = rng.random() < 0.4
    eol = b"\r\n" if rng.random() < 0.3 else b"\n"
    if rng.random() < 0.3:
        out += b"leading junk" + eol
    for r in range(int(rng.integers(0, 12))):
        seq = bytes(rng.choice(alpha, size=int(rng.integers(0, 90))))
        name = b"r%d" % r + (b" comment" if rng.random() < 0.5 else b"")
        if fastq:
            out += b"@" + name + eol + seq + eol + b"+" + eol
            q = b"I" * len(seq)
            if rng.random() < 0.1:
                q = q[:-1]                       # truncated quality -> stream ends
            out += q + eol
        else:
            out += b">" + name + eol
            w = int(rng.integers(1, 40))
            for j in range(0, len(seq), w):
                line = seq[j:j + w]
                if rng.random() < 0.05:
                    line = line[:1] + b">@+" + line[1:]   # header chars inside a line are data
                out += line + eol
                if rng.random() < 0.1:
                    out += eol                    # blank line
    if rng.random() < 0.3 and out.endswith(eol):
        out = out[: -len(eol)]
    return bytes(out)


def test_reader_fuzz_against_oracle_reader(tmp_path):
    rng = np.random.default_rng(77)
    p = str(tmp_path / "f.fx")
    for trial in range(300):
        blob = _random_fastx(rng)
        with open(p, "wb") as f:
            f.write(blob)
        s, o = device.read_all(p)
        os_, oo = orc.fastx_read(p)
        assert np.array_equal(o, oo), (trial, blob)
        assert np.array_equal(s[: int(o[-1])], os_[: int(oo[-1])]), (trial, blob)


def _collect(reader):
    reads = []
    for s, o in reader:
        reads += orc.reads_of(s, o)
    return reads


@pytest.mark.parametrize("name", ["edge.fasta", "edge_crlf.fasta", "edge.fastq", "edge.fa.gz", "weird.fasta"])
@pytest.mark.parametrize("chunk", [64, 1000, 1 << 20])
def test_parallel_reader_matches_oracle_on_fixtures(name, chunk):
    exp = orc.reads_of(*orc.fastx_read(golden_path(name)))
    with device.ParallelReader(golden_path(name), threads=4, chunk_bytes=chunk) as rd:
        assert _collect(rd) == exp


def test_parallel_reader_fuzz_small_chunks(tmp_path):
    """Range cuts at every possible place: tiny chunks on fuzzed FASTA.  Files with a '+'
    line must be refused (LRB_ERR_FORMAT), never parsed differently."""
    from lrbinner_amd._lib import LrbError
    rng = np.random.default_rng(99)
    p = str(tmp_path / "f.fa")
    parsed = refused = 0
    for trial in range(300):
        blob = _random_fastx(rng)
        with open(p, "wb") as f:
            f.write(blob)
        exp = orc.reads_of(*orc.fastx_read(p))
        try:
            with device.ParallelReader(p, threads=3, chunk_bytes=int(rng.integers(64, 200))) as rd:
                got = _collect(rd)
        except LrbError as e:
            assert e.code == 6 and b"+" in blob
            refused += 1
            continue
        assert got == exp, (trial, blob)
        parsed += 1
    assert parsed > 200


def test_cli_flags_and_defaults():
    import lrbinner
    p = lrbinner.build_parser()
    a = p.parse_args(["reads", "-r", "x.fasta", "-o", "out"])
    assert (a.k_size, a.bin_size, a.bin_count, a.ae_epochs, a.ae_dims, a.ae_hidden, a.threads) == \
        (3, 10, 32, 200, 8, "128,128", 8)                    # lrbinner.py:15-58
    assert (a.min_bin_size, a.bin_iterations, a.separate, a.cuda, a.resume) == (10000, 1000, False, False, False)
    a = p.parse_args(["reads", "-r", "x.fq", "-o", "o", "-k", "4", "-bs", "32", "-bc", "10", "--ae-dims", "4",
                      "-mbs", "5000", "-bit", "0", "-t", "32", "--cuda", "--resume", "-sep"])
    assert (a.k_size, a.bin_size, a.bin_count, a.ae_dims, a.min_bin_size, a.bin_iterations, a.threads) == \
        (4, 32, 10, 4, 5000, 0, 32) and a.cuda and a.resume and a.separate
    a = p.parse_args(["contigs", "-r", "x.fa", "-c", "c.fa", "-o", "o"])
    assert a.mode == "contigs" and a.contigs == "c.fa"
    with pytest.raises(SystemExit):
        p.parse_args(["reads", "-r", "x.fa", "-o", "o", "-k", "6"])   # k limited to 3..5


def test_cli_rejects_unknown_extension_and_missing_file(tmp_path):
    import lrbinner
    with pytest.raises(SystemExit) as e:
        lrbinner.main(["reads", "-r", str(tmp_path / "reads.txt"), "-o", str(tmp_path / "o1")])
    assert e.value.code == 1
    with pytest.raises(SystemExit) as e:
        lrbinner.main(["reads", "-r", str(tmp_path / "missing.fasta"), "-o", str(tmp_path / "o2")])
    assert e.value.code == 1
    # kernel limits are reported before any stage runs, not after the VAE has trained
    fa = tmp_path / "r.fasta"
    fa.write_text(">a\nACGT\n")
    for extra in (["--ae-dims", "128"], ["-bc", "2000"], ["-bs", "0"]):
        with pytest.raises(SystemExit) as e:
            lrbinner.main(["reads", "-r", str(fa), "-o", str(tmp_path / "o3")] + extra)
        assert e.value.code == 1
        assert not os.path.exists(str(tmp_path / "o3" / "profiles" / "com_profs"))


def test_library_shuffle_is_random_shuffle():
    """lrb_mt_shuffle_i64: the same permutation as random.shuffle from the same generator
    state, and the generator left where random.shuffle leaves it (the draws after it agree)."""
    import random
    from lrbinner_amd.device import py_shuffle
    for seed, n in ((1, 0), (2, 1), (3, 2), (4, 3), (5, 1000), (6, 70_001)):
        random.seed(seed)
        for _ in range(seed * 37):      # start somewhere inside the 624-word block
            random.random()
        state = random.getstate()
        want = list(range(n))
        random.shuffle(want)
        follow = [random.random(), random.getrandbits(17), random.randrange(1000)]
        random.setstate(state)
        got = py_shuffle(np.arange(n))
        assert got.tolist() == want
        assert [random.random(), random.getrandbits(17), random.randrange(1000)] == follow


def test_npy_cache_hands_out_only_what_is_on_disk(tmp_path):
    """_npcache: np.save + remember; load returns the remembered array only while the file is
    the one this process wrote, otherwise reads the file."""
    import os
    from lrbinner_amd import _npcache
    a = np.arange(12, dtype=np.float64).reshape(3, 4)
    p = str(tmp_path / "x")
    _npcache.save(p, a)
    assert os.path.exists(p + ".npy")
    assert _npcache.load(p + ".npy") is a and _npcache.load(p) is a
    b = a * 2
    np.save(p, b)                                   # somebody else rewrote the file
    os.utime(p + ".npy", ns=(1, 1))
    got = _npcache.load(p + ".npy")
    assert got is not a and np.array_equal(got, b)
    _npcache.save(p, a)
    _npcache.drop(p)
    assert _npcache.load(p) is not a
    with pytest.raises(OSError):
        _npcache.load(str(tmp_path / "missing.npy"))


def test_q6_to_values_is_float_of_the_token_for_every_q():
    """The side-car path of stage 3_1: q / 1e6 (torch's threaded division or numpy's) is the
    double float("d.dddddd") gives, for all 10^6 + 1 six-decimal values."""
    from lrbinner_amd import runners_utils as ru
    q = np.arange(0, 1_000_001, dtype=np.uint32)
    got = ru.q6_to_values(q)
    want = q.astype(np.float64) / 1e6
    assert got.dtype == np.float64 and np.array_equal(got.view(np.uint64), want.view(np.uint64))
    for v in (0, 1, 36341, 99999, 100000, 333333, 500000, 999999, 1000000):
        assert got[v] == float("%d.%06d" % (v // 1000000, v % 1000000))
    rng = np.random.default_rng(1)
    pick = rng.integers(0, 1_000_001, 20000)
    assert all(got[v] == float("%d.%06d" % (v // 1000000, v % 1000000)) for v in pick)


def test_contig_records_are_walked_once(tmp_path, monkeypatch):
    """contig_records: the first walk parses, later walks (fragmenting, output) reuse ids and
    sequences; a changed file is parsed again; want_seqs=False hands out ids only."""
    import os
    from lrbinner_amd import runners_utils as ru
    p = str(tmp_path / "c.fasta")
    open(p, "w").write(">a desc\nACGT\nAC\n>b\n\n>c\nTTTT\n")
    calls = []

    class Counting(ru._NativeContigs):       # every parse of a file is one of these (lrb_fasta_scan)
        def __init__(self, path):
            calls.append(path)
            super().__init__(path)

    monkeypatch.setattr(ru, "_NativeContigs", Counting)
    first = list(ru.contig_records(p))
    assert first == [("a", b"ACGTAC"), ("b", b""), ("c", b"TTTT")] and len(calls) == 1
    del calls[:]
    assert list(ru.contig_records(p)) == first and calls == []                     # from memory
    assert list(ru.contig_records(p, want_seqs=False)) == [("a", None), ("b", None), ("c", None)]
    open(p, "w").write(">z\nGG\n")
    os.utime(p, ns=(5, 5))
    assert list(ru.contig_records(p)) == [("z", b"GG")] and len(calls) == 1       # the file changed
    ru.release_contigs(p)
    assert list(ru.contig_records(p)) == [("z", b"GG")] and len(calls) == 2
    ru.release_contigs()


def test_contig_votes_follow_reference_order():
    """cluster_utils.py:496-515: candidates are collected cluster by cluster (labels in order of
    first appearance), so a tied contig goes to the label that appeared first in the file, and
    bins.txt lists contigs in the order that walk first meets them."""
    from lrbinner_amd.pipelines import contig_votes
    # fragment 0 (contig Z) makes label 0 the first cluster; contig A is tied 2 : 2
    labels = np.array([0, 1, 1, 0, 0, -1, 2])
    parent = {0: "Z", 1: "A", 2: "A", 3: "A", 4: "A", 5: "N", 6: "B"}
    got = contig_votes(labels, parent)
    assert got == {"Z": 0, "A": 0, "B": 2}
    assert list(got) == ["Z", "A", "B"]          # cluster 0: Z, A; cluster 1: A again; cluster 2: B
    # without Z ahead of it the same contig goes to label 1 (first label in the file)
    got = contig_votes(labels[1:], {i - 1: p for i, p in parent.items() if i})
    assert got == {"A": 1, "B": 2} and list(got) == ["A", "B"]
    # a clear majority is a majority whatever the order
    assert contig_votes([3, 5, 5, 5, -1], {0: "c", 1: "c", 2: "c", 3: "c", 4: "c"}) == {"c": 5}


def test_stage_with_late_artifact_reruns_without_purging(tmp_path):
    """A stage whose file appears after its checkpoint (the deferred 15-mer table file): logged +
    file missing => run again on resume, later stages keep their checkpoints."""
    from lrbinner_amd import pipelines as P
    cp = ru.Checkpointer(str(tmp_path / "ck"))
    art = str(tmp_path / "table")
    ran = []
    for stage, params in (("1_1", ["r", 3]), ("1_2", ["r"]), ("2_1", ["r", 10, 32]), ("3_1", ["numpy"])):
        P._stage(cp, stage, params, "s", "d", "k", lambda s=stage: ran.append(s), artifact=art if stage == "1_2" else None)
    assert ran == ["1_1", "1_2", "2_1", "3_1"]
    cp2 = ru.Checkpointer(str(tmp_path / "ck"), True)
    ran.clear()
    P._stage(cp2, "1_2", ["r"], "s", "d", "k", lambda: ran.append("1_2"), artifact=art)   # file never appeared
    assert ran == ["1_2"] and sorted(cp2.completed) == ["1_1", "1_2", "2_1", "3_1"]
    open(art, "w").close()
    P._stage(cp2, "1_2", ["r"], "s", "d", "k", lambda: ran.append("again"), artifact=art)
    assert ran == ["1_2"]
    # changed parameters: the reference's rule (log purges the later majors)
    P._stage(cp2, "1_2", ["other"], "s", "d", "k", lambda: ran.append("changed"), artifact=art)
    assert ran[-1] == "changed" and sorted(cp2.completed) == ["1_1", "1_2"]


def test_sim8_stage_isolation_scores_on_record():
    """tests/golden/e2e_reference_8g.json (make_golden_sim8.py, build container): the reference's own
    pipeline on the 8-genome stand-in, >= 3 seeds -- and stage isolation (ii): latents trained by THIS
    build's fused HIP trainer on the GPU box, clustered by the REFERENCE's perform_binning in the
    container.  Their median F1 is within 0.5 of the reference-trained
    ones' and the median number of bins is the same: the VAE stage is not where a defect could hide."""
    ref = json.load(open(golden_path("e2e_reference_8g.json")))
    assert len(ref["runs"]) >= 3 and ref["n_reads"] == 40350
    assert ref["reference_latents_reclustered"][0]["f1"] == ref["runs"][0]["f1"]   # clustering is deterministic given the latent
    hip = ref["hip_latents_reference_clustering"]
    assert len(hip) >= 3
    assert abs(np.median([r["f1"] for r in hip]) - np.median([r["f1"] for r in ref["runs"]])) <= 0.5
    assert np.median([r["bins"] for r in hip]) == ref["bins_median"]
    # the block-mixture data of round 1's Sim-8-scale run: the reference splits those genomes too
    blk = json.load(open(golden_path("e2e_reference_blocks.json")))
    assert all(r["bins"] > 8 and r["recall"] < 70 for r in blk["runs"])


def test_native_contig_parse_equals_the_line_loop(tmp_path, monkeypatch):
    """lrb_fasta_scan + lrb_fasta_write_fragments (one native pass over the contigs file) against the Python line
    loop they replace: same records (ids, sequences), same fragments file, same contig -> fragments and fragment ->
    contig maps -- wrapped lines, CRLF, blank lines, junk before the first header, an empty id, duplicate ids,
    internal blanks, no newline at the end; plain and gzip.  Both are held to Bio.SeqIO's FASTA record semantics,
    which is what the reference reads contigs with (SimpleFastaParser: title = line[1:].rstrip(), id = its first
    word, sequence = "".join(line.rstrip() for the other lines).replace(" ", "").replace("\r", "")): blocks of ten
    separated by blanks and carriage returns inside a line vanish, a leading tab stays."""
    import gzip

    def bio_records(blob):
        recs, title, lines = [], None, []
        for line in re.split(b"\r\n|\r|\n", blob):          # text mode with universal newlines: a lone '\r' ends a line
            if line[:1] == b">":
                if title is not None:
                    recs.append((title, b"".join(lines).replace(b" ", b"").replace(b"\r", b"")))
                words = line[1:].rstrip().split(None, 1)
                title, lines = (words[0].decode() if words else ""), []
            elif title is not None:
                lines.append(line.rstrip())
        if title is not None:
            recs.append((title, b"".join(lines).replace(b" ", b"").replace(b"\r", b"")))
        return recs

    from lrbinner_amd import runners_utils as ru
    rng = np.random.default_rng(4)
    parts = [b"; not a record\nACGT\n"]
    for i in range(300):
        L = int(rng.choice([0, 1, 59, 60, 61, 2499, 2500, 4999, 5000, 5001, 7500, 12345]))
        seq = bytes(rng.choice(list(b"ACGTNacgt"), L).astype(np.uint8))
        name = b"" if i == 17 else (b"dup" if i % 50 == 3 else b"ctg%d" % i)
        nl = b"\r\n" if i % 7 == 0 else b"\n"
        parts.append(b">" + name + (b" len=%d\t x" % L if i % 3 else b"") + nl)
        wrap = [60, 80, 10 ** 9][i % 3]
        for a in range(0, L, wrap):
            parts.append((b"  " if i % 11 == 0 else b"") + seq[a:a + wrap] + nl)
        if i % 13 == 0:
            parts.append(nl)
    parts.append(b">blocks\nACGTACGTAC GTACGTACGT ACGTA\n  ACGT\rACGT \r\n\tGGCC\n")
    # old-Mac line ends: a tab before a lone '\r' is trailing white space of its line, a '>' behind one opens a record
    parts.append(b">cr\nACG\t\rTTT\r>after_cr extra\rGGG\rCC\n")
    parts.append(b">last\nAC GT")
    blob = b"".join(parts)
    plain, gz = str(tmp_path / "c.fasta"), str(tmp_path / "c.fasta.gz")
    open(plain, "wb").write(blob)
    with gzip.open(gz, "wb") as f:
        f.write(blob)
    for path in (plain, gz):
        res = {}
        for native in ("0", "1"):
            monkeypatch.setenv("LRB_CONTIGS_NATIVE", native)
            ru.release_contigs()
            recs = list(ru.contig_records(path))
            assert (ru._contigs_native(path) is not None) == (native == "1")
            out = tmp_path / f"o{native}{os.path.basename(path)}"
            os.makedirs(out / "fragments")
            groups, parent = ru.split_contigs(path, str(out))
            ids, lens = ru.contig_lengths(path)
            res[native] = (recs, dict(groups), parent, open(out / "fragments" / "contigs.fasta", "rb").read(), ids, lens)
            # a second walk comes from the cache
            assert list(ru.contig_records(path)) == recs
            assert [c for c, s in ru.contig_records(path, want_seqs=False)] == ids
        assert res["0"] == res["1"]
        assert len(res["1"][0]) == 304 and res["1"][0][-1] == ("last", b"ACGT")
        assert res["1"][0][-4] == ("blocks", b"ACGTACGTACGTACGTACGTACGTAACGTACGT\tGGCC")
        assert res["1"][0][-3:-1] == [("cr", b"ACGTTT"), ("after_cr", b"GGGCC")]
        assert res["1"][0] == bio_records(blob)
    ru.release_contigs()


def test_contig_votes_vectorised_equals_the_reference_walk():
    """contig_votes against a line-by-line restatement of cluster_utils.py:496-515 on random labelings: ties,
    noise-only contigs, contigs whose fragments are not contiguous, duplicated ids, one cluster, no cluster."""
    from collections import Counter, defaultdict
    from lrbinner_amd.pipelines import contig_votes

    def walk(labels, parent):
        clusters = defaultdict(list)
        for frag, lab in enumerate(labels):
            if lab != -1:
                clusters[lab].append(frag)
        pc = defaultdict(list)
        for lab, frags in clusters.items():
            for frag in frags:
                pc[parent[frag]].append(lab)
        return {c: Counter(v).most_common()[0][0] for c, v in pc.items()}

    rng = np.random.default_rng(21)
    for trial in range(60):
        n_frag = int(rng.integers(1, 400))
        n_lab = int(rng.integers(1, 7))
        labels = rng.integers(-1, n_lab, n_frag) if trial % 5 else np.full(n_frag, -1 if trial % 10 else 3)
        n_cont = int(rng.integers(1, 40))
        owner = np.sort(rng.integers(0, n_cont, n_frag)) if trial % 3 else rng.integers(0, n_cont, n_frag)
        parent = {i: f"c{int(owner[i]) % 7 if trial % 4 == 0 else int(owner[i])}" for i in range(n_frag)}
        want = walk(labels.tolist(), parent)
        got = contig_votes(labels, parent)
        assert got == want and list(got) == list(want), trial
        assert all(type(v) is int for v in got.values())


def test_recorded_reference_search_on_this_builds_latents_agrees_seed_by_seed():
    """tests/golden/e2e_recluster_big.json (ref_recluster.py, build container): the REFERENCE's cluster_points on a
    latent.npy this build trained for the 432 k-read stand-in, under random.seed(1..8), next to this build's search
    on the same latents under the same seeds (GPU box, scripts/e2e_merge_probe.py).  On record: identical cluster
    sizes for every seed -- including the seeds under which both merge two genomes."""
    import json
    d = json.load(open(golden_path("e2e_recluster_big.json")))
    # ... and the other way round: both searches on the REFERENCE's own latents of that data set (one of eight seeds merges)
    for side, bins in (("on_this_builds_latents", [7, 7, 7, 7, 8, 8, 8, 8]), ("on_the_references_latents", [7, 8, 8, 8, 8, 8, 8, 8])):
        runs = d[side]["runs"]
        assert len(runs) == 8
        for r in runs:
            assert r["reference_cluster_sizes"] == r["this_build_cluster_sizes"], (side, r["seed"])
            assert sum(r["reference_cluster_sizes"]) > 420_000
        assert sorted(r["reference_bins"] for r in runs) == bins


def test_multi_gpu_profile_stages_route_and_checkpoints(tmp_path, monkeypatch):
    """LRB_GPUS=N: stages 1_1 / 1_2 / 2_1 go to ONE child job of N ranks (lrbinner_amd.dist) and are logged with the
    reference's stage ids and parameters (pipelines.py:269-302), so that --resume skips them; a resume that needs
    only some of them stays on one GPU; without LRB_GPUS nothing changes."""
    from lrbinner_amd import pipelines as P
    from lrbinner_amd import dist as ld
    out = tmp_path / "o"
    (out / "profiles").mkdir(parents=True)
    calls = []

    def fake_spawn(n, argv, module="lrbinner_amd.dist"):
        calls.append((n, [str(a) for a in argv]))
        for f in ("com_profs", "cov_profs", "15mers-counts"):
            (out / "profiles" / f).write_text("x")
        return 0

    monkeypatch.setattr(ld, "spawn_ranks", fake_spawn)
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.delenv("LRB_GPUS", raising=False)
    cp = ru.Checkpointer(str(out / "checkpoints"))
    assert P.gpus_requested() == (None, 1)
    assert P._sharded_profile_stages(cp, "r.fasta", str(out), 4, 10, 32, 8) is False and not calls
    monkeypatch.setenv("LRB_GPUS", "8")
    assert P.gpus_requested() == ("spawn", 8)
    assert P._sharded_profile_stages(cp, "r.fasta", str(out), 4, 10, 32, 8) is True
    assert calls == [(8, ["--reads", "r.fasta", "--output", str(out), "-k", "4", "-bs", "10", "-bc", "32", "-t", "8"])]
    assert cp.completed == {"1_1": ["r.fasta", 4], "1_2": ["r.fasta"], "2_1": ["r.fasta", 10, 32]}
    # resume: nothing is due, nothing is started
    cp2 = ru.Checkpointer(str(out / "checkpoints"), True)
    assert P._sharded_profile_stages(cp2, "r.fasta", str(out), 4, 10, 32, 8) is False and len(calls) == 1
    # another bin size: only 2_1 is due -> the single-GPU stage function takes it
    assert P._sharded_profile_stages(cp2, "r.fasta", str(out), 4, 32, 10, 8) is False and len(calls) == 1
    # a failing child ends the run with its status, as check_proc does for the reference's binaries
    monkeypatch.setattr(ld, "spawn_ranks", lambda *a, **k: 3)
    with pytest.raises(SystemExit) as e:
        P._sharded_profile_stages(ru.Checkpointer(str(out / "ck2")), "r.fasta", str(out), 4, 10, 32, 8)
    assert e.value.code == 3


def test_bench_gpus_flag_starts_that_many_ranks(monkeypatch):
    """`python bench.py --gpus N` (the driver's command) starts N ranks as a child job before anything touches the
    GPU and leaves with the child's status; under a launcher the world has to be the one asked for."""
    import subprocess
    import sys
    import types
    import bench
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=7)

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "5", "--warmup", "2"])
    monkeypatch.delenv("RANK", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.launch_ranks(types.SimpleNamespace(gpus=4))
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "5", "--warmup", "2"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # one GPU: nothing is started
    seen.clear()
    bench.launch_ranks(types.SimpleNamespace(gpus=1))
    assert not seen
    # under a launcher: the same world or a refusal
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "4")
    bench.launch_ranks(types.SimpleNamespace(gpus=4))
    with pytest.raises(SystemExit) as e:
        bench.launch_ranks(types.SimpleNamespace(gpus=8))
    assert e.value.code == 2
    assert not seen


def test_npy_files_written_in_the_background_follow_the_late_artifact_rule(tmp_path):
    """Stage 3_1 hands its arrays on in memory and lets writer threads put the .npy files on the disk
    (_npcache.save_async): the file appears under its name only when complete, load() serves the array meanwhile,
    and a run that died between 3_1 and the end of the write finds 3_1 logged but its files missing on --resume:
    the stage runs again and the later checkpoints stay (pipelines.py:309-330; same rule as 15mers-counts)."""
    from lrbinner_amd import _npcache
    from lrbinner_amd import pipelines as P
    a = np.arange(12, dtype=np.float64).reshape(3, 4)
    path = str(tmp_path / "com_profs.npy")
    _npcache.save_async(path, a)
    assert _npcache.load(path) is a                       # before or after the write: the same array
    _npcache.finish()
    assert os.path.exists(path) and not os.path.exists(path + ".part")
    assert np.array_equal(np.load(path), a) and _npcache.load(path) is a
    os.utime(path, ns=(1, 1))                             # somebody else's file now
    assert _npcache.load(path) is not a
    _npcache.drop()
    # a write that fails is reported by finish(), and nothing appears under the name
    bad = str(tmp_path / "no_such_dir" / "x.npy")
    _npcache.save_async(bad, a)
    with pytest.raises(OSError):
        _npcache.finish()
    assert not os.path.exists(bad)
    # the checkpoint rule with two late artifacts
    cp = ru.Checkpointer(str(tmp_path / "ck"))
    arts = [str(tmp_path / "a.npy"), str(tmp_path / "b.npy")]
    ran = []
    P._stage(cp, "3_1", ["numpy"], "s", "d", "k", lambda: ran.append("3_1"), artifact=arts)
    P._stage(cp, "4_1", ["o", 8], "s", "d", "k", lambda: ran.append("4_1"))
    cp2 = ru.Checkpointer(str(tmp_path / "ck"), True)     # the run died before the files were complete
    open(arts[0], "w").close()
    P._stage(cp2, "3_1", ["numpy"], "s", "d", "k", lambda: ran.append("3_1 again"), artifact=arts)
    assert ran == ["3_1", "4_1", "3_1 again"] and sorted(cp2.completed) == ["3_1", "4_1"]
    open(arts[1], "w").close()
    P._stage(cp2, "3_1", ["numpy"], "s", "d", "k", lambda: ran.append("no"), artifact=arts)
    assert ran[-1] == "3_1 again"


def test_com_profs_converted_early_is_what_the_stage_would_have_made(tmp_path):
    """pipelines._convert_early starts the float64 conversion of com_profs behind the k-mer stage; stage 3_1 picks
    the array up, writes the same .npy as an in-line conversion, and nothing is started when the stage is not due or
    the side-car is missing (pipelines.py:315-321: the values are float(token) of the text either way)."""
    from lrbinner_amd import _npcache
    from lrbinner_amd import pipelines as P
    out = str(tmp_path)
    os.makedirs(f"{out}/profiles")
    rng = np.random.default_rng(5)
    q = {"com_profs": rng.integers(0, 10 ** 6 + 1, size=(40, 32), dtype=np.uint32),
         "cov_profs": rng.integers(0, 10 ** 6 + 1, size=(40, 10), dtype=np.uint32)}
    for name, v in q.items():
        path = f"{out}/profiles/{name}"
        with open(path, "w") as f:
            for row in v:
                f.write(" ".join("%.6f" % (int(x) / 1e6) for x in row) + " \n")
        side = ru._ValueSidecar(path)
        side.append(v)
        side.close()
    cp = ru.Checkpointer(str(tmp_path / "ck"))
    P._convert_early(cp, "3_1", out, "com_profs")
    key = os.path.abspath(f"{out}/profiles/com_profs")
    assert key in P._early
    P._early[key][0].join()
    early = P._early[key][1]["arr"]
    P._stage(cp, "3_1", ["numpy"], "s", "d", "k", lambda: P._profiles_to_npy(out), artifact=P._npy_artifacts(out))
    assert key not in P._early
    assert _npcache.load(f"{out}/profiles/com_profs.npy") is early          # the array made early is the one handed on
    _npcache.finish()
    for name, v in q.items():
        got = np.load(f"{out}/profiles/{name}.npy")
        text = np.array([[float(t) for t in line.split()] for line in open(f"{out}/profiles/{name}")])
        assert got.dtype == np.float64 and np.array_equal(got, text)
    # the stage is logged and its files are there: nothing to start
    P._convert_early(cp, "3_1", out, "com_profs")
    assert key not in P._early
    # due again (a file is gone) but no side-car: left to the stage (the text parse holds the GIL)
    os.remove(f"{out}/profiles/com_profs.npy")
    os.remove(f"{out}/profiles/com_profs.q6.json")
    P._convert_early(cp, "3_1", out, "com_profs")
    assert key not in P._early
    _npcache.drop()


def _fake_kfd(tmp_path, n_gpus, cpu_nodes=1):
    """A KFD topology tree as /sys/class/kfd/kfd/topology/nodes shows it: CPU nodes first (simd_count 0), then GPUs."""
    root = tmp_path / "nodes"
    for i in range(cpu_nodes + n_gpus):
        d = root / str(i)
        d.mkdir(parents=True)
        gpu = i >= cpu_nodes
        (d / "properties").write_text(f"cpu_cores_count {0 if gpu else 64}\nsimd_count {1024 if gpu else 0}\n"
                                      f"location_id {((5 + i) << 8) if gpu else 0}\ndomain 0\n")
    return str(root)


def test_gpu_count_from_the_kfd_topology(tmp_path, monkeypatch):
    """_gpus.visible_gpus: GPU nodes of the KFD topology, narrowed by the *_VISIBLE_DEVICES variables; None when the
    topology cannot be read (then nothing is refused)."""
    from lrbinner_amd import _gpus
    root = _fake_kfd(tmp_path, 8, cpu_nodes=2)
    assert len(_gpus.kfd_gpus(root)) == 8 and _gpus.kfd_gpus(root)[0]["bdf"] == "0000:07:00.0"
    assert _gpus.visible_gpus(root, env={}) == 8
    assert _gpus.visible_gpus(root, env={"HIP_VISIBLE_DEVICES": "0,3"}) == 2
    assert _gpus.visible_gpus(root, env={"ROCR_VISIBLE_DEVICES": "1", "HIP_VISIBLE_DEVICES": "0,1"}) == 1
    assert _gpus.visible_gpus(str(tmp_path / "nothing"), env={}) is None
    assert _gpus.pin_to_gpu_numa(0, root=str(tmp_path / "nothing"))["pinned"] is False
    assert _gpus.pin_to_gpu_numa(3, root=root)["pinned"] is False          # (no NUMA file for the fake devices)
    # HIP device i is KFD GPU i only while no *_VISIBLE_DEVICES variable reorders or subsets the node
    assert _gpus.kfd_index_of(3, 8, env={}) == 3
    assert _gpus.kfd_index_of(1, 8, env={"HIP_VISIBLE_DEVICES": "4,5,6,7"}) == 5
    assert _gpus.kfd_index_of(0, 8, env={"ROCR_VISIBLE_DEVICES": "2,3", "HIP_VISIBLE_DEVICES": "1"}) == 3
    assert _gpus.kfd_index_of(2, 8, env={"HIP_VISIBLE_DEVICES": "4,5"}) is None
    assert _gpus.kfd_index_of(0, 8, env={"ROCR_VISIBLE_DEVICES": "GPU-abcdef"}) is None


def test_bench_refuses_more_gpus_than_the_node_shows(tmp_path, monkeypatch, capsys):
    """`python bench.py --gpus 8` on a node with fewer GPUs: exit 2 and a message BEFORE anything is started (a
    rendezvous of 8 ranks on 1 GPU would hang until the driver's timeout); with the GPUs there, the child job is
    started with torchrun's own rendezvous port."""
    import subprocess
    import sys
    import types
    import bench
    from lrbinner_amd import _gpus
    started = []
    monkeypatch.setattr(subprocess, "run", lambda cmd, env=None, **kw: started.append(cmd) or types.SimpleNamespace(returncode=0))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.delenv("LRB_BENCH_BACKEND", raising=False)
    monkeypatch.setattr(_gpus, "visible_gpus", lambda *a, **k: 1)
    with pytest.raises(SystemExit) as e:
        bench.launch_ranks(types.SimpleNamespace(gpus=8))
    assert e.value.code == 2 and not started and "shows 1 GPU" in capsys.readouterr().err
    monkeypatch.setattr(_gpus, "visible_gpus", lambda *a, **k: 8)
    with pytest.raises(SystemExit) as e:
        bench.launch_ranks(types.SimpleNamespace(gpus=8))
    assert e.value.code == 0 and len(started) == 1
    cmd = started[0]
    assert "--standalone" in cmd and cmd[cmd.index("--local-addr") + 1] == "127.0.0.1" and "--master-port" not in cmd
    # the rehearsal backend (several ranks on one GPU through gloo) is not refused
    started.clear()
    monkeypatch.setenv("LRB_BENCH_BACKEND", "gloo")
    monkeypatch.setattr(_gpus, "visible_gpus", lambda *a, **k: 1)
    with pytest.raises(SystemExit) as e:
        bench.launch_ranks(types.SimpleNamespace(gpus=2))
    assert e.value.code == 0 and len(started) == 1


def test_spawn_ranks_refuses_and_times_out(monkeypatch, capsys):
    """dist.spawn_ranks (LRB_GPUS=N lrbinner.py reads): the same refusal; and the collective timeout of the process
    group is minutes (LRB_COLLECTIVE_TIMEOUT_S), not torch's default."""
    import subprocess
    import types
    from lrbinner_amd import _gpus
    from lrbinner_amd import dist as ld
    started = []
    monkeypatch.setattr(subprocess, "run", lambda cmd, env=None, **kw: started.append(cmd) or types.SimpleNamespace(returncode=0))
    monkeypatch.delenv("LRB_DIST_BACKEND", raising=False)
    monkeypatch.setattr(_gpus, "visible_gpus", lambda *a, **k: 2)
    assert ld.spawn_ranks(8, ["--reads", "r.fa", "--output", "o"]) == 2 and not started
    assert "shows 2 GPU" in capsys.readouterr().err
    assert ld.spawn_ranks(2, ["--reads", "r.fa", "--output", "o"]) == 0 and "--standalone" in started[0]
    monkeypatch.delenv("LRB_COLLECTIVE_TIMEOUT_S", raising=False)
    assert ld.collective_timeout().total_seconds() == 600
    monkeypatch.setenv("LRB_COLLECTIVE_TIMEOUT_S", "45")
    assert ld.collective_timeout().total_seconds() == 45


def test_host_pack_is_the_device_layout_bit_for_bit(tmp_path):
    """lrb_pack_reads_host -- what the parser pool runs per range so that 0.375 bytes a base cross PCIe instead of 1 --
    writes the packed HBM layout of lrb_pack_layout / pack_kernel: against the numpy model of that layout (helpers.np_pack,
    which the GPU tests hold the pack kernel to) on the reference's edge files and on random bytes incl. lowercase, N, CR
    and header characters; the AVX2 / BMI2 path and the scalar loop alike; and the pool's packed view of every batch of a
    file == the pack of that batch's ASCII view."""
    from helpers import np_pack, golden_path
    from oracle import oracle as orc
    from lrbinner_amd import device as lrb
    rng = np.random.default_rng(1)
    alphabet = np.frombuffer(b"ACGTNacgtn\r>@x", np.uint8)
    reads = [bytes(rng.choice(alphabet, size=int(rng.integers(0, 400)))) for _ in range(400)] + [b"", b"A", b"ACGT" * 8, b"C" * 33]
    sets = [orc.fastx_read(golden_path("edge.fasta")), orc.fastx_read(golden_path("weird.fasta")), orc.concat(reads)]
    for buf, offs in sets:
        codes, mask, co, mo = np_pack(buf, offs)[:4]
        for scalar in (False, True):
            hp = lrb.pack_reads_host(buf, offs, scalar=scalar)
            assert np.array_equal(hp.codes, codes) and np.array_equal(hp.mask, mask), scalar
            assert np.array_equal(hp.code_off, co) and np.array_equal(hp.mask_off, mo)
            assert np.array_equal(hp.lens[:hp.n], np.diff(offs).astype(np.uint32))
    # a slice of a batch (offsets that do not start at 0)
    buf, offs = sets[2]
    sub = np.ascontiguousarray(offs[100:301])
    hp = lrb.pack_reads_host(buf, sub)
    codes, mask, co, mo = np_pack(buf[int(sub[0]):int(sub[-1])], sub - sub[0])[:4]
    assert np.array_equal(hp.codes, codes) and np.array_equal(hp.mask, mask)
    # the pool's packed view
    fa = tmp_path / "reads.fa"
    clean = [r.replace(b">", b"A").replace(b"@", b"C").replace(b"\r", b"G") for r in reads]
    fa.write_bytes(b"".join(b">r%d\n" % i + r + b"\n" for i, r in enumerate(clean)))
    seen = 0
    with lrb.ParallelReader(str(fa), threads=3, chunk_bytes=7000, packed=True) as rd:
        assert rd.parallel
        while True:
            hp = rd.next_packed()
            if hp is None:
                break
            want = lrb.pack_reads_host(np.concatenate([np.frombuffer(r, np.uint8) for r in clean[seen:seen + hp.n]] + [np.zeros(0, np.uint8)]),
                                       np.concatenate([[0], np.cumsum([len(r) for r in clean[seen:seen + hp.n]])]).astype(np.uint64))
            assert np.array_equal(hp.codes[:len(want.codes)], want.codes) and np.array_equal(hp.mask[:len(want.mask)], want.mask)
            assert np.array_equal(hp.code_off, want.code_off) and np.array_equal(hp.mask_off, want.mask_off)
            seen += hp.n
    assert seen == len(clean)


def test_bench_line_is_compact_and_carries_every_stage(tmp_path, monkeypatch):
    """bench.compact_line (round 6): the contract line stays below 8 KB and carries, INSIDE `roofline`, every stage's
    fraction / kernel time / traffic ratio and the C4-shaped rank -- rebuilt here from the detail file of the recorded run
    (profiles/r06_bench_detail.json) and compared with the recorded line."""
    import importlib.util
    from helpers import ROOT
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rec = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_detail.json")))
    full = dict(rec["line"])
    full.update({k: v for k, v in rec["detail"].items() if k in ("extra", "roofline_stages", "vae_step", "c4_phases", "cpu_baseline")})
    full["roofline"] = dict(rec["detail"]["roofline"])
    monkeypatch.setenv("LRB_BENCH_DETAIL", str(tmp_path / "detail.json"))
    line = bench.compact_line(full)
    text = json.dumps(line)
    assert len(text) < 8192, len(text)
    assert not {"extra", "roofline_stages", "vae_step", "c4_phases"} & set(line)
    st = line["roofline"]["stages"]
    assert {"k1_k4", "k1_k5", "k2", "k3_default", "k3_kept_lists", "k4_seed_hist", "k5_gauss", "vae_encode", "vae_encode_c3", "k6_core", "k6_mst",
            "k3_bins64_default", "k3_bins64_kept_lists", "vae_step_c1_b1024", "vae_step_c3_b8192"} <= set(st)
    for name, e in st.items():
        assert set(e) == {"frac", "kernel_ms", "traffic_ratio", "bound"} and e["frac"] > 0 and e["kernel_ms"] > 0, name
    assert st == rec["line"]["roofline"]["stages"]
    # fraction = algorithmic bytes / time / 8 TB/s, recomputable from the line alone (3,044 B a read at k = 4)
    assert st["k1_k4"]["frac"] == pytest.approx(3044 * 1_000_000 / (st["k1_k4"]["kernel_ms"] * 1e-3) / 8e12, rel=2e-3)
    assert st["k2"]["frac"] == pytest.approx(82388 * 400_000 / (st["k2"]["kernel_ms"] * 1e-3) / 8e12, rel=2e-3)
    cr = line["roofline"]["c4_rank"]
    assert cr["world_size_seen_by_rccl"] == 1 and cr["default_reads_per_s"] > 1e7 and cr["kept_reads_per_s"] > cr["default_reads_per_s"]
    assert cr["default_reads_per_s"] == pytest.approx(2_500_000 / (cr["phases_ms_max_over_ranks"]["total_ms"] * 1e-3), rel=2e-3)
    assert json.load(open(tmp_path / "detail.json"))["detail"]["c4_phases"]["routes"]["default"]["reads_per_s"] > 0


def test_batch_groups_are_filled_from_the_end():
    """runners_utils._batch_groups (the rule of lrb_packed_group_starts): consecutive batches, at most max_bases a group (one
    batch at least), filled FROM THE END -- the last group, whose window lists the table stage leaves in the workspaces, is a
    full one; the remainder is the FIRST group.  C1's 432,333 reads of 10 kb in batches of 6,705: 32 k + 400 k, not 400 k + 32 k."""
    class B:
        def __init__(self, n):
            self.total_bases = n
    rng = np.random.default_rng(5)
    for trial in range(200):
        sizes = rng.integers(1, 100, size=int(rng.integers(0, 40))).tolist()
        if trial % 7 == 0 and sizes:
            sizes[int(rng.integers(0, len(sizes)))] = 500          # a batch larger than the limit: a group of its own
        bs = [B(x) for x in sizes]
        groups = list(ru._batch_groups(bs, 150))
        assert [b for g, _ in groups for b in g] == bs                         # consecutive, in order, nothing lost
        for g, bases in groups:
            assert bases == sum(b.total_bases for b in g) and (bases <= 150 or len(g) == 1)
        for i in range(1, len(groups)):                                        # every group but the first is full:
            assert groups[i][1] + groups[i - 1][0][-1].total_bases > 150         # the batch in front of it would not fit
    per = 6705 * 10_000
    bs = [B(per)] * 64 + [B(432_333 * 10_000 - 64 * per)]
    g = list(ru._batch_groups(bs, 4_000_000_000))
    assert len(g) == 2 and g[0][1] < 400_000_000 and g[1][1] > 3_900_000_000
    # (any measure of a batch: the sharded driver's items are (batch id, packed) pairs)
    items = [(i, B(60)) for i in range(7)]
    assert [len(x) for x, _ in ru._batch_groups(items, 150, bases_of=lambda it: it[1].total_bases)] == [1, 2, 2, 2]


def test_profile_writer_places_a_later_group_first(tmp_path):
    """runners_utils._ProfileWriter.seek_rows (round 6): the coverage stage formats the rows of its LAST group first (that
    group's window lists are still in the workspaces) and the other groups afterwards; rows are fixed-width, so the pair
    (text, six-decimal integers) of a group belongs at row x row_bytes / row x 4 cols whatever the order it arrives in.  The
    file and its side-car must be what an in-order writer produces, the side-car's description included."""
    row_bytes, cols = 9 * 3, 3
    rows = [(b"%08d " % i) * 2 + b"%08d\n" % i for i in range(50)]
    assert all(len(r) == row_bytes for r in rows)
    q = np.arange(50 * cols, dtype=np.uint32).reshape(50, cols)

    def write(order):
        path = str(tmp_path / ("cov_" + "_".join(map(str, order[0]))))
        with open(path, "wb") as out:
            side = ru._ValueSidecar(path)
            wr = ru._ProfileWriter(out, side)
            try:
                for a, b in order:
                    wr.seek_rows(a, row_bytes, cols)
                    for s0 in range(a, b, 7):          # several pairs a group, as the chunked formatter hands them over
                        s1 = min(b, s0 + 7)
                        slot = wr.slot()
                        wr.put(slot, np.frombuffer(b"".join(rows[s0:s1]), np.uint8), q[s0:s1])
                wr.seek_rows(50, row_bytes, cols)
            finally:
                wr.close()
            out.flush()
            side.close()
        meta = json.load(open(path + ".q6.json"))
        return open(path, "rb").read(), open(path + ".q6", "rb").read(), meta

    want = write([(0, 50)])
    assert want[0] == b"".join(rows) and want[1] == q.tobytes() and want[2] == {"cols": cols, "rows": 50, "text_bytes": 50 * row_bytes}
    got = write([(37, 50), (0, 12), (12, 37)])          # the last group first, then the others in order
    assert got == want
    assert ru.load_value_sidecar is not None


def test_c1_hard_outcomes_by_class_against_the_reference():
    """The recorded WHOLE runs on the hard set (no GPU work here; made by make_golden_sim8.py and
    scripts/r06_accuracy_runs.py), split by WHAT merged -- round 5 lumped every run below eight bins:
      strain  genomes 6 and 7 (the 10 %-diverged strain pair) in one bin, F1 97.2;
      gc      genomes 5 and 7 (GC 0.535 / 0.57) in one bin, F1 92.3;
      both    six bins, F1 89.2.
    One one-sided Fisher exact test per class (is this build's rate above the reference's?), the table printed.  None may
    reject at 1 %.  These binary samples have little power -- round 5's 25 reference runs showed no strain merge, which a
    5.5 % rate gives one time in four; the runs made since show it (seeds 32, 39, 45: the sample is extended in the build
    container all round, scripts/r06_ref_streams.sh) -- the comparison that has power is the next two tests.  Also kept: every 8-bin run of this build within +-0.5 F1 of a reference
    8-bin run, their means within +-0.1, every merged run at the F1 its class costs."""
    from helpers import hard_set_statistics, _outcome_class
    st = hard_set_statistics()
    print({k: v for k, v in st.items() if not k.endswith("_runs") and k != "classes"})
    for cls, row in st["classes"].items():
        print(f"  {cls:7s} reference {row['ref']:2d} of {row['n_ref']}   this build {row['build']:2d} of {row['n_build']} (p = {row['fisher_p_build_worse']:.3f})"
              f"   with round 5's runs {row['build_r5_and_r6']} of {row['n_build_r5_and_r6']} (p = {row['fisher_p_build_worse_r5_and_r6']:.3f})")
        assert row["fisher_p_build_worse"] >= 0.01 and row["fisher_p_build_worse_r5_and_r6"] >= 0.01, (cls, row)
    assert st["n_ref"] >= 25 and st["n_b"] >= 100
    ref8 = [r["f1"] for r in st["ref_runs"] if r["bins"] >= 8]
    our8 = [r["f1"] for r in st["our_runs"] if r["bins"] >= 8]
    assert abs(np.mean(our8) - np.mean(ref8)) <= 0.1
    for r in our8:
        assert min(abs(r - f) for f in ref8) <= 0.5
    cost = {"strain": (96.9, 97.4), "gc": (92.0, 92.6), "both": (88.9, 89.5)}
    for r in st["our_runs"] + st["ref_runs"]:
        cls = _outcome_class(r)
        assert cls in ("none", "strain", "gc", "both"), r
        if cls != "none":
            assert cost[cls][0] <= r["f1"] <= cost[cls][1], (cls, r["f1"])
    # the same seed, the same candidate order (random.seed(s) before the search on either side): seeds 1-25
    ours25 = sum(r["bins"] < 8 for r in st["our_runs"] if r["seed"] <= 25)
    ref25 = sum(r["bins"] < 8 for r in st["ref_runs"] if r["seed"] <= 25)
    print(f"  seeds 1-25: reference {ref25} runs below eight bins, this build {ours25}")


def test_c1_hard_latent_separation_is_the_references():
    """The CONTINUOUS statistic the round-5 verdict asked for: helpers.latent_pair_stats of every recorded run's latent.npy
    -- per genome pair d' = angle between the two centroids / pooled RMS angle of the reads to their own centroid, in the
    unit-sphere geometry the cluster search works in (cluster_utils.py:31-42) -- for the strain pair (6, 7) and the GC pair
    that merges (5, 7), reference runs against this build's 100, two-sided Mann-Whitney.  Stated power (simulated with the
    measured spread, sigma 0.42): with 25 reference runs a shift of 0.4 in d' (6 %) is found 93 times in 100 at the 1 % level,
    with 50 a shift of 0.3 is found 92 times in 100.  Measured: strain 6.756 +- 0.354 (reference) against 6.729 +- 0.463,
    p = 0.86; (5, 7) 6.581 against 6.570, p = 0.96.  Asserted: no rejection at 1 %, medians within 0.25 of each other."""
    from scipy.stats import mannwhitneyu
    from helpers import hard_set_statistics
    st = hard_set_statistics()
    ref = [r for r in st["ref_runs"] if "pairs" in r]
    ours = [r for r in st["our_runs"] if "pairs" in r]
    assert len(ref) >= 25 and len(ours) >= 100
    for pair in ("strain", "gc57", "gc56"):
        x = np.array([r["pairs"][pair]["dprime"] for r in ref])
        y = np.array([r["pairs"][pair]["dprime"] for r in ours])
        p = mannwhitneyu(x, y, alternative="two-sided").pvalue
        print(f"  d'({pair}): reference {x.mean():.3f} +- {x.std():.3f} (n = {len(x)})   this build {y.mean():.3f} +- {y.std():.3f} (n = {len(y)})   Mann-Whitney p = {p:.3f}")
        assert p >= 0.01, (pair, p)
        assert abs(np.median(x) - np.median(y)) <= 0.25 * (1 + (pair == "gc56")), (pair, np.median(x), np.median(y))
    # a merged run is not a run with closer latents: the merge is the search's (next test)
    merged = np.array([r["pairs"]["strain"]["dprime"] for r in ours if [6, 7] in r["merged"]])
    whole = np.array([r["pairs"]["strain"]["dprime"] for r in ours if not r["merged"]])
    print(f"  this build, strain d' of the runs that merged the pair {merged.mean():.3f} (n = {len(merged)}) against {whole.mean():.3f} of the 8-bin runs")
    assert abs(merged.mean() - whole.mean()) <= 0.5


def test_c1_hard_mergeability_under_equal_search_seeds():
    """Where the 7-bin runs come from, and whether this build's latents make more of them.  A whole run's outcome is a draw
    of the cluster SEARCH: its candidates are tried in an order random.seed fixes, and a candidate between two genomes
    yields a cluster that holds both.  So the same latent.npy clustered under other search seeds merges or not, seed by
    seed, and some seeds are bad for everybody (seed 6: 9 of 25 reference latents, 9 of 40 of this build's).  Compared
    under EQUAL seeds, all through this build's search (which finds the reference's clusters seed for seed:
    tests/test_gpu_sim8.py::test_sim8_reference_latents_through_this_clustering, and the own-seed check below):
      * per search (seeds 1-8): the reference's latents (25 of session 2, 51 with the runs made since: 42 of 408 searches
        merge a pair = 10.3 %, the strain pair in 28), this build's fused step 33 of 320 = 10.3 % (24), its torch-module path
        23 of 320 (16); seeds 1001-1003: 16 of 81 against 66 of 300;
      * per latent, the share of its eight searches that merge: Mann-Whitney, no rejection at 1 %;
      * the first-step statistic (scripts/r06_accuracy_runs.first_step: of 400 random candidates on the full matrix, the
        share of accepted ones whose cluster holds more than half of two genomes -- continuous, one value per latent):
        reference 0.0415 +- 0.024 (n = 51), fused 0.0461 +- 0.031, torch modules 0.0422 +- 0.028, p = 0.65 / 0.94.  Power: at
        these spreads and sample sizes a rise of 0.02 (a build whose latents are half again as mergeable) is found 4 times in
        5 at the 1 % level.
    The reference's first 25 whole runs showing no strain merge was what its own latents' 5.5 % per search gives one time
    in four (0.945 ** 25 = 0.24); its runs 26-50 show it (tests/golden/e2e_reference_c1_hard.json)."""
    from scipy.stats import mannwhitneyu
    from helpers import ROOT
    prof = lambda name: json.load(open(os.path.join(ROOT, "profiles", name)))
    ref18 = prof("r06_c1hard_ref_recluster_s1to8.json")["latents"]
    ref1001 = prof("r06_c1hard_ref_recluster_s1001.json")["latents"]
    more = os.path.join(ROOT, "profiles", "r06_c1hard_ref_recluster_s1to8_more.json")
    ref_fs = list(ref1001)          # the latents that carry the first-step statistic
    if os.path.exists(more):        # the reference runs made later in the round (26 ...): the same seeds 1-8
        extra = json.load(open(more))["latents"]
        have = {it["file"] for it in ref18}
        ref18 = ref18 + [it for it in extra if it["file"] not in have]
        have = {it["file"] for it in ref_fs}
        ref_fs = ref_fs + [it for it in extra if it["file"] not in have]
    fused = prof("r06_c1hard_runs_s1to8.json")["runs"]
    torchp = prof("r06_c1hard_runs_torch_s1to8.json")["runs"]
    fused1001 = prof("r06_c1hard_runs_100.json")["runs"]
    assert [q["search_seed"] for q in ref18[0]["searches"]] == [q["search_seed"] for q in fused[0]["searches"]] == list(range(1, 9))
    assert [q["search_seed"] for q in ref1001[0]["searches"]] == [q["search_seed"] for q in fused1001[0]["searches"]] == [1001, 1002, 1003]

    def per_search(items):
        return sum(bool(q["merged"]) for it in items for q in it["searches"]), sum(len(it["searches"]) for it in items)

    def per_latent(items):
        return np.array([np.mean([bool(q["merged"]) for q in it["searches"]]) for it in items])

    from helpers import fisher_one_sided
    for name, a, b in (("seeds 1-8, fused step", ref18, fused), ("seeds 1-8, torch modules", ref18, torchp), ("seeds 1001-1003, fused step", ref1001, fused1001)):
        (kr, nr), (kb, nb) = per_search(a), per_search(b)
        p_l = mannwhitneyu(per_latent(a), per_latent(b), alternative="two-sided").pvalue
        print(f"  {name}: reference latents {kr} of {nr} searches merge a pair, this build's {kb} of {nb}; per-latent shares Mann-Whitney p = {p_l:.3f}")
        assert p_l >= 0.01
        assert kb / nb <= kr / nr + 0.06          # (per-search rates: within six points of the reference latents' own)
    fs = lambda items, k_: np.array([it["first_step"][k_] for it in items])
    for k_ in ("merged_rate", "strain_rate"):
        x = fs(ref_fs, k_)
        for name, items in (("fused step", fused), ("torch modules", torchp)):
            y = fs(items, k_)
            p = mannwhitneyu(x, y, alternative="two-sided").pvalue
            print(f"  first-step {k_}: reference {x.mean():.4f} +- {x.std():.4f} (n = {len(x)})   {name} {y.mean():.4f} +- {y.std():.4f} (n = {len(y)})   p = {p:.3f}")
            assert p >= 0.01 and abs(x.mean() - y.mean()) <= 0.015, (k_, name, p, x.mean(), y.mean())
    # bad seeds are bad for everybody: the seed that merges most reference latents merges most of this build's too
    worst_ref = int(np.argmax([sum(bool(it["searches"][j]["merged"]) for it in ref18[:25]) for j in range(8)]))
    worst_ours = int(np.argmax([sum(bool(it["searches"][j]["merged"]) for it in fused) for j in range(8)]))
    assert worst_ref == worst_ours == 5        # search seed 6


def test_c1_plain_recorded_outcomes_against_the_reference():
    """BASELINE config C1's own stand-in (helpers.synth_sim8_c1, README.md:73's flags), recorded samples, no GPU work: the
    reference's whole runs (tests/golden/e2e_reference_c1.json; extended all round in the build container) against 40 of
    this build (profiles/r06_c1_runs_40.json): runs below eight bins by one-sided Fisher (no rejection at 1 %), every 8-bin run
    of this build within +-0.5 F1 of a reference run and the means within +-0.1 (north_star's tolerance); and under EQUAL
    search seeds 1-8 the reference's latents against this build's (profiles/r06_c1_ref_recluster_s1to8.json,
    r06_c1_runs_s1to8.json): per-search merge rate within three points, the first-step mergeability statistic by
    Mann-Whitney (no rejection at 1 %).  On this set merges are rare on both sides (a few per cent of searches)."""
    from scipy.stats import mannwhitneyu
    from helpers import ROOT, fisher_one_sided
    ref = json.load(open(golden_path("e2e_reference_c1.json")))["runs"]
    ours = json.load(open(os.path.join(ROOT, "profiles", "r06_c1_runs_40.json")))["runs"]
    assert len(ref) >= 9 and len(ours) == 40
    k_ref, k_b = sum(r["bins"] < 8 for r in ref), sum(r["bins"] < 8 for r in ours)
    p = fisher_one_sided(k_b, len(ours), k_ref, len(ref))
    print(f"  C1 stand-in: reference {k_ref} of {len(ref)} runs below eight bins, this build {k_b} of {len(ours)} (one-sided Fisher p = {p:.3f})")
    assert p >= 0.01
    ref8 = [r["f1"] for r in ref if r["bins"] >= 8]
    our8 = [r["f1"] for r in ours if r["bins"] >= 8]
    assert abs(np.mean(our8) - np.mean(ref8)) <= 0.1
    assert all(min(abs(f - g) for g in ref8) <= 0.5 for f in our8)
    rl = json.load(open(os.path.join(ROOT, "profiles", "r06_c1_ref_recluster_s1to8.json")))["latents"]
    more = os.path.join(ROOT, "profiles", "r06_c1_ref_recluster_s1to8_more.json")
    if os.path.exists(more):     # the reference runs made later in the round
        rl = rl + [it for it in json.load(open(more))["latents"] if it["file"] not in {q["file"] for q in rl}]
    bl = json.load(open(os.path.join(ROOT, "profiles", "r06_c1_runs_s1to8.json")))["runs"]
    rate = lambda items: sum(bool(q["merged"]) for it in items for q in it["searches"]) / sum(len(it["searches"]) for it in items)
    x = np.array([it["first_step"]["merged_rate"] for it in rl])
    y = np.array([it["first_step"]["merged_rate"] for it in bl])
    pm = mannwhitneyu(x, y, alternative="two-sided").pvalue
    print(f"  seeds 1-8: reference latents ({len(rl)}) merge a pair in {100 * rate(rl):.1f} % of searches, this build's (40) in {100 * rate(bl):.1f} %; "
          f"first-step {x.mean():.4f} against {y.mean():.4f}, Mann-Whitney p = {pm:.3f}")
    assert rate(bl) <= rate(rl) + 0.03 and pm >= 0.01


def test_cpu_budget_follows_the_cgroup_quota(monkeypatch):
    """_gpus.cpu_budget: the affinity mask cut by the cgroup's quota (cpu.max of v2, cfs_quota / cfs_period of v1): the
    parser pool and the host packer are sized from it (a GPU box of the pool shows 256 CPUs and grants the time of 16)."""
    import builtins, io
    from lrbinner_amd import _gpus, runners_utils as ru
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(256)))
    files = {}
    real_open = builtins.open

    def fake_open(path, *a, **k):
        if isinstance(path, str) and path.startswith("/sys/fs/cgroup/"):
            if path in files:
                return io.StringIO(files[path])
            raise FileNotFoundError(path)
        return real_open(path, *a, **k)
    monkeypatch.setattr(builtins, "open", fake_open)
    assert _gpus.cpu_budget() == 256                                   # no cgroup files: the affinity mask
    files["/sys/fs/cgroup/cpu.max"] = "max 100000\n"
    assert _gpus.cpu_budget() == 256
    files["/sys/fs/cgroup/cpu.max"] = "1600000 100000\n"
    assert _gpus.cpu_budget() == 16
    assert ru.parser_threads(64) == 8 and ru.parser_threads(4) == 4    # half the budget, at least four, at most what was asked
    monkeypatch.delenv("LRB_HOST_PACK", raising=False)
    assert ru.host_packs() is False                                    # below 32 CPUs the packer would compete with the parser
    files["/sys/fs/cgroup/cpu.max"] = "6400000 100000\n"
    assert _gpus.cpu_budget() == 64 and ru.host_packs() is True and ru.parser_threads(64) == 32
    del files["/sys/fs/cgroup/cpu.max"]
    files["/sys/fs/cgroup/cpu/cpu.cfs_quota_us"] = "800000\n"
    files["/sys/fs/cgroup/cpu/cpu.cfs_period_us"] = "100000\n"
    assert _gpus.cpu_budget() == 8
    files["/sys/fs/cgroup/cpu/cpu.cfs_quota_us"] = "-1\n"
    assert _gpus.cpu_budget() == 256
    monkeypatch.setenv("LRB_HOST_PACK", "1")
    assert ru.host_packs() is True
    monkeypatch.setenv("LRB_HOST_PACK", "0")
    assert ru.host_packs() is False
