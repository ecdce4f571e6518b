"""CPU-only checks of the C ABI: the library loads and exports every symbol of
include/lrb_hip.h, and its host-side pieces (reader, text rows, LUT, layout)
agree with the oracle and with the reference-generated fixtures.  No kernels run."""
import os
import re

import numpy as np
import pytest

from helpers import GOLDEN, golden_path, gz_bytes, parse_profile_text, random_reads
from oracle import oracle as orc

from lrbinner_amd import _lib, device


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(os.path.dirname(GOLDEN), "..", "include", "lrb_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(lrb_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    L = _lib.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"liblrb_hip.so does not export {name}"
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "SO_PATH", "/nonexistent/liblrb_hip.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.lib()


@pytest.mark.parametrize("k", [3, 4, 5])
def test_lut_matches_oracle(k):
    lut, dim = device.kmer_lut(k)
    olut, odim = orc.kmer_lut(k)
    assert dim == odim and np.array_equal(lut, olut)
    assert device.kmer_dim(k) == dim


def test_bad_k_is_an_error_code_not_a_crash():
    with pytest.raises(_lib.LrbError) as e:
        device.kmer_dim(7)
    assert e.value.code == 1


def test_pack_layout_properties():
    offs = np.array([0, 0, 1, 16, 33, 10033], dtype=np.uint64)
    lens, co, mo = device.pack_layout(offs)
    assert lens.tolist() == [0, 1, 15, 17, 10000]
    for r, L in enumerate(lens):
        cw = int(co[r + 1] - co[r])
        mw = int(mo[r + 1] - mo[r])
        assert cw % 4 == 0 and mw % 4 == 0 and co[r] % 4 == 0
        assert cw >= -(-int(L) // 16) + 4 and mw >= -(-int(L) // 32) + 4


@pytest.mark.parametrize("name", ["edge.fasta", "edge_crlf.fasta", "edge.fastq", "edge.fa.gz",
                                  "weird.fasta"])
def test_reader_matches_oracle_reader(name):
    s, o = device.read_all(golden_path(name))
    os_, oo = orc.fastx_read(golden_path(name))
    assert np.array_equal(o, oo)
    assert np.array_equal(s[: int(o[-1])], os_[: int(oo[-1])])


def test_reader_batches_concatenate(tmp_path):
    rng = np.random.default_rng(5)
    reads = random_reads(rng, 57, 0, 400, p_n=0.02)
    p = tmp_path / "r.fa"
    with open(p, "wb") as f:
        for i, r in enumerate(reads):
            f.write(b">r%d\n" % i + r + b"\n")
    got = []
    with device.FastxReader(str(p)) as rd:
        while True:
            b = rd.next_batch(max_reads=10)
            if b is None:
                break
            s, o = b
            assert len(o) - 1 <= 10
            got += orc.reads_of(s, o)
    assert got == reads


def test_reader_truncated_fastq_ends_stream(tmp_path):
    p = tmp_path / "t.fq"
    p.write_bytes(b"@a\nACGT\n+\nIIII\n@b\nACGTAC\n+\nII\n")  # second record: short quality
    s, o = device.read_all(str(p))
    os_, oo = orc.fastx_read(str(p))
    assert np.array_equal(o, oo) and len(o) - 1 == 1


def test_reader_missing_file():
    with pytest.raises(_lib.LrbError) as e:
        device.FastxReader("/nonexistent/x.fa")
    assert e.value.code == 5


@pytest.mark.parametrize("k", [3, 4, 5])
def test_com_text_from_integer_counts_matches_reference(k):
    buf, offs = orc.fastx_read(golden_path("edge.fasta"))
    counts, _ = orc.count_kmers(buf, offs, k)
    lens = np.diff(offs).astype(np.uint32)
    txt, vals = device.format_com(counts, lens, k, threads=3, want_values=True)
    gold = gz_bytes(f"com_profs_k{k}.txt.gz")
    assert txt == gold
    # the values are what pipelines.py:315-321 parses out of the text
    assert np.array_equal(vals, parse_profile_text(gold))


@pytest.mark.parametrize("bs,bc", [(10, 32), (32, 10), (4, 10)])
def test_cov_text_from_integer_hist_matches_reference(bs, bc):
    buf, offs = orc.fastx_read(golden_path("edge.fasta"))
    keys, cnts = orc.k15_sparse(buf, offs)
    hist, sums = orc.cov_hist(buf, offs, keys, cnts, bs, bc)
    txt, vals = device.format_cov(hist, sums.astype(np.uint32), threads=2, want_values=True)
    gold = gz_bytes(f"cov_profs_bs{bs}_bc{bc}.txt.gz")
    assert txt == gold
    assert np.array_equal(vals, parse_profile_text(gold))


def test_format_zero_rows():
    txt = device.format_com(np.zeros((0, 32), np.uint32), np.zeros(0, np.uint32), 3)
    assert txt == b""


def test_percent_f_formatter_is_libc_exact():
    """The library prints "%f" with integer arithmetic; it must agree with libc on every
    value: all count/total quotients the path can produce are of that form, plus ties,
    subnormals, large values and the snprintf fallbacks."""
    import ctypes as C
    L = _lib.lib()
    a, b = C.create_string_buffer(64), C.create_string_buffer(64)

    def same(v):
        L.lrb_debug_format_f(float(v), a, b)
        assert a.value == b.value, (v, a.value, b.value)

    rng = np.random.default_rng(1)
    for t in list(range(1, 400)) + [9998, 9997, 9986, 32986, 123457]:
        for c in {0, 1, 2, 3, t // 7, t // 3, t // 2, t - 1, t} | set(rng.integers(0, t + 1, 12).tolist()):
            same(c / t)
    for v in rng.random(20000):
        same(v)
    for v in (rng.random(5000) * 10.0 ** rng.integers(-12, 9, 5000)):
        same(v)
    for v in (0.0, 1.0, 0.5e-6, 1.5e-6, 2.5e-6, 0.0000005, 0.0000015, 0.1234565, 0.1234575, 5e-324, 1e-300,
              123456.7890125, 2.0 ** 39, 2.0 ** 40, 1e15, float("inf"), float("nan"), -0.25, -0.0):
        same(v)
    k = np.arange(0, 2 ** 21, dtype=np.float64) * 2.0 ** -21   # exact binary fractions: many exact ties
    for v in k[::37]:
        same(v)


def test_header_is_plain_c(tmp_path):
    """include/lrb_hip.h is what a C (or cgo / JNI) binding compiles against: it must be valid C99
    on its own, and a C program must link against the library through it."""
    import subprocess
    from helpers import ROOT
    src = tmp_path / "use.c"
    src.write_text('#include "lrb_hip.h"\n#include <stdio.h>\n'
                   'int main(void) { unsigned dim = 0; int rc = lrb_kmer_dim(4, &dim);\n'
                   '  printf("%d %u %d\\n", rc, dim, lrb_version()); return rc; }\n')
    inc = os.path.join(ROOT, "include")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-fsyntax-only", f"-I{inc}", str(src)], check=True)
    exe = tmp_path / "use"
    libdir = os.path.join(ROOT, "lrbinner_amd")
    subprocess.run(["gcc", "-std=c99", f"-I{inc}", str(src), "-o", str(exe), f"-L{libdir}", "-llrb_hip",
                    f"-Wl,-rpath,{libdir}"], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert out[0] == "0" and out[1] == "136"


def test_drop_in_executables_are_built_and_parse_their_argv():
    """lrbinner_amd/bin/{count-kmers,count-15mers,search-15mers} (built by lrbinner_amd/csrc/Makefile from thin
    mains over the C ABI): present, linked against the in-tree library, and -- like the reference's binaries
    started with too few arguments -- exit non-zero before touching a device (no GPU here)."""
    import subprocess
    from helpers import ROOT
    for name, few in (("count-kmers", ["reads.fa", "out"]), ("count-15mers", ["reads.fa"]),
                      ("search-15mers", ["table", "reads.fa", "out", "10"])):
        exe = os.path.join(ROOT, "lrbinner_amd", "bin", name)
        assert os.access(exe, os.X_OK), exe
        r = subprocess.run([exe] + few, capture_output=True, text=True)
        assert r.returncode != 0 and "usage" in r.stderr
        ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
        assert "lrbinner_amd/bin/../liblrb_hip.so" in ldd or "liblrb_hip.so =>" in ldd
