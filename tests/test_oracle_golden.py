"""The CPU oracle (oracle/lrb_oracle.c) against the fixtures generated from the
REAL reference binaries (tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN, golden_path, gz_bytes
from oracle import oracle as orc


@pytest.fixture(scope="module")
def edge():
    return orc.fastx_read(golden_path("edge.fasta"))


def test_meta_table_file_layout():
    meta = json.load(open(golden_path("meta.json")))
    # kmer_utils.h:89-97 -- u64 entry count then 4^15 u32
    assert meta["table_header_word"] == 4 ** 15
    assert meta["table_file_bytes"] == 8 + 4 * 4 ** 15


def test_lut_known_answers():
    # SURVEY 8a row 3 (probed on the reference): k=3 canonical numbering
    lut, D = orc.kmer_lut(3)
    assert D == 32
    code = {"A": 0, "C": 1, "T": 2, "G": 3}
    idx = lambda s: int(lut[code[s[0]] * 16 + code[s[1]] * 4 + code[s[2]]])
    known = {"AAA": 0, "AAC": 1, "AAT": 2, "AAG": 3, "ACA": 4, "ACC": 5, "ACT": 6, "ACG": 7,
             "ATA": 8, "ATC": 9, "ATG": 10, "AGA": 11, "AGC": 12, "AGG": 13, "CAA": 14,
             "CAC": 15, "CAG": 16, "CCA": 17, "CCC": 18, "CCG": 19, "CTA": 20, "CTC": 21,
             "CGA": 22, "CGC": 23, "TAA": 24, "TAC": 25, "TCA": 26, "TCC": 27, "TTC": 28,
             "TGC": 29, "GAC": 30, "GCC": 31}
    for s, v in known.items():
        assert idx(s) == v, s
    assert idx("GAA") == idx("TTC") and idx("GCA") == idx("TGC") and idx("TTT") == idx("AAA")
    assert orc.kmer_lut(4)[1] == 136 and orc.kmer_lut(5)[1] == 512


def test_revcomp():
    # ACG (0,1,3) -> CGT (1,3,2)
    assert orc.revcomp(0b000111, 3) == 0b011110
    for k in (3, 4, 5, 15):
        rng = np.random.default_rng(k)
        for x in rng.integers(0, 4 ** k, size=50):
            assert orc.revcomp(orc.revcomp(int(x), k), k) == int(x)
    # k odd: no k-mer is its own reverse complement
    assert all(orc.revcomp(x, 3) != x for x in range(64))


@pytest.mark.parametrize("k", [3, 4, 5])
def test_composition_text_matches_reference(edge, k):
    buf, offs = edge
    counts, totals = orc.count_kmers(buf, offs, k)
    txt = orc.format_com(orc.com_profile(counts, totals))
    assert txt == gz_bytes(f"com_profs_k{k}.txt.gz")


@pytest.mark.parametrize("name", ["edge_crlf.fasta", "edge.fastq", "edge.fa.gz"])
def test_reader_format_invariance(edge, name):
    buf, offs = orc.fastx_read(golden_path(name))
    assert np.array_equal(offs, edge[1])
    assert np.array_equal(buf[: int(offs[-1])], edge[0][: int(offs[-1])])


def test_sparse_table_matches_reference(edge):
    g = np.load(golden_path("k15_sparse.npz"))
    keys, cnts = orc.k15_sparse(*edge)
    assert np.array_equal(keys, g["idx"])
    assert np.array_equal(cnts, g["cnt"])


@pytest.mark.parametrize("bs,bc", [(10, 32), (32, 10), (4, 10)])
def test_coverage_text_matches_reference(edge, bs, bc):
    buf, offs = edge
    keys, cnts = orc.k15_sparse(buf, offs)
    hist, sums = orc.cov_hist(buf, offs, keys, cnts, bs, bc)
    txt = orc.format_cov(orc.cov_profile(hist, sums))
    assert txt == gz_bytes(f"cov_profs_bs{bs}_bc{bc}.txt.gz")


def test_weird_headers_and_lines():
    buf, offs = orc.fastx_read(golden_path("weird.fasta"))
    reads = orc.reads_of(buf, offs)
    assert len(reads) == 8
    assert reads[1] == b"ACGT@ACGT>ACGT+ACGTACGTACGTAAAC"
    assert reads[5] == b""
    counts, totals = orc.count_kmers(buf, offs, 3)
    assert orc.format_com(orc.com_profile(counts, totals)) == gz_bytes("weird_com_k3.txt.gz")
    keys, cnts = orc.k15_sparse(buf, offs)
    g = np.load(golden_path("weird_k15_sparse.npz"))
    assert np.array_equal(keys, g["idx"]) and np.array_equal(cnts, g["cnt"])
    hist, sums = orc.cov_hist(buf, offs, keys, cnts, 4, 10)
    assert orc.format_cov(orc.cov_profile(hist, sums)) == gz_bytes("weird_cov_bs4_bc10.txt.gz")


def test_coverage_bin_probe_values():
    # SURVEY appendix B, probed on the reference: bin_size=4, bins=10
    exp = {1: 0, 2: 0, 4: 0, 5: 9, 6: 9, 7: 9, 8: 1, 12: 2, 40: 9, 41: 9, 100: 9, 0: 0}
    for c, b in exp.items():
        assert orc.cov_bin(c, 4, 10) == b, c


def test_missing_file_raises():
    with pytest.raises(OSError):
        orc.fastx_read(os.path.join(GOLDEN, "does_not_exist.fa"))
