"""The process + file boundary itself: lrbinner_amd/bin/{count-kmers,count-15mers,search-15mers} are drop-ins
for the reference's three executables (same argv -- count-kmers.cpp:195-198, count-15mers.cpp:101-103,
search-15mers.cpp:124-136 --, same output files, exit status as the only signal), started here the way
mbcclr_utils/runners_utils.py:78-105 starts them (os.system) and held to the files the REAL reference
binaries wrote for the same inputs (tests/golden, make_golden.py)."""
import os

import numpy as np
import pytest

from helpers import ROOT, golden_path, gz_bytes

pytestmark = pytest.mark.gpu

BIN = os.path.join(ROOT, "lrbinner_amd", "bin")


def _system(cmd):
    return os.system(cmd)


@pytest.fixture(scope="module")
def table_file(tmp_path_factory):
    d = tmp_path_factory.mktemp("bins")
    out = str(d / "15mers-counts")
    # runners_utils.py:88-95
    assert _system(f'"{BIN}/count-15mers" "{golden_path("edge.fasta")}" "{out}" 8') == 0
    yield out
    os.remove(out)


@pytest.mark.parametrize("name", ["edge.fasta", "edge_crlf.fasta", "edge.fastq", "edge.fa.gz"])
@pytest.mark.parametrize("k", [3, 4, 5])
def test_count_kmers_writes_the_reference_file(tmp_path, name, k):
    out = str(tmp_path / "com_profs")
    # runners_utils.py:78-85: "{bin}/count-kmers" "{reads}" "{output}/profiles/com_profs" {k} {threads}
    assert _system(f'"{BIN}/count-kmers" "{golden_path(name)}" "{out}" {k} 8 > /dev/null') == 0
    assert open(out, "rb").read() == gz_bytes(f"com_profs_k{k}.txt.gz")


def test_count_15mers_writes_the_reference_table(table_file):
    assert os.path.getsize(table_file) == 8 + 4 * 4 ** 15
    with open(table_file, "rb") as f:
        assert int(np.frombuffer(f.read(8), dtype="<u8")[0]) == 4 ** 15
    t = np.memmap(table_file, dtype="<u4", mode="r", offset=8)
    g = np.load(golden_path("k15_sparse.npz"))           # non-zero (index, count) pairs of the reference's file
    assert np.array_equal(t[g["idx"].astype(np.int64)], g["cnt"])
    # nothing else is set: the sum over the file is the sum over the pairs
    total = 0
    for a in range(0, 4 ** 15, 1 << 26):
        total += int(t[a:a + (1 << 26)].sum(dtype=np.uint64))
    assert total == int(g["cnt"].sum(dtype=np.uint64))


@pytest.mark.parametrize("bs,bc", [(10, 32), (32, 10), (4, 10)])
def test_search_15mers_writes_the_reference_file(tmp_path, table_file, bs, bc):
    out = str(tmp_path / "cov_profs")
    # runners_utils.py:98-105: "{bin}/search-15mers" "{output}/profiles/15mers-counts" "{reads}" "{out}" {bin_size} {bins} {threads}
    assert _system(f'"{BIN}/search-15mers" "{table_file}" "{golden_path("edge.fastq")}" "{out}" {bs} {bc} 8') == 0
    assert open(out, "rb").read() == gz_bytes(f"cov_profs_bs{bs}_bc{bc}.txt.gz")


def test_failures_are_exit_codes(tmp_path):
    """check_proc (runners_utils.py:108-113) sees a non-zero status: missing input, bad k, unwritable output."""
    out = str(tmp_path / "o")
    assert _system(f'"{BIN}/count-kmers" "{tmp_path}/missing.fasta" "{out}" 3 8 2> /dev/null') != 0
    assert _system(f'"{BIN}/count-kmers" "{golden_path("edge.fasta")}" "{out}" 9 8 2> /dev/null') != 0
    assert _system(f'"{BIN}/count-kmers" "{golden_path("edge.fasta")}" "{tmp_path}/no/such/dir/o" 3 8 2> /dev/null') != 0
    assert _system(f'"{BIN}/search-15mers" "{tmp_path}/missing-table" "{golden_path("edge.fasta")}" "{out}" 10 32 8 2> /dev/null') != 0
    assert _system(f'"{BIN}/count-15mers" "{golden_path("edge.fasta")}" 2> /dev/null') != 0     # too few arguments
