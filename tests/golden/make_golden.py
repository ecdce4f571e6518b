#!/usr/bin/env python3
"""Generate the committed golden fixtures from the REAL reference binaries.

Run in the build container only (needs oracle/_ref, i.e. /root/reference):

    make -C oracle && python tests/golden/make_golden.py

Inputs are synthetic edge-case files written by this script (our own data);
expected outputs are what the reference binaries (count-kmers, count-15mers,
search-15mers compiled by oracle/Makefile from the reference sources where
they lie) print for them.  Only inputs + expected outputs are stored here --
no reference source text.

Writes (all under tests/golden/):
  edge.fasta, edge_crlf.fasta, edge.fastq, edge.fa.gz, weird.fasta
  com_profs_k{3,4,5}.txt.gz          count-kmers output for edge.fasta
  cov_profs_bs{B}_bc{C}.txt.gz       search-15mers output, (B,C) in (10,32),(32,10),(4,10)
  k15_sparse.npz                     non-zero (index,count) pairs of 15mers-counts
  weird_com_k3.txt.gz, weird_cov_bs4_bc10.txt.gz, weird_k15_sparse.npz
  meta.json                          sizes, header word of the table file, sha256 of outputs
"""
import gzip
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref")


def rand_seq(rng, n, alphabet=b"ACGT"):
    return bytes(rng.choice(np.frombuffer(alphabet, dtype=np.uint8), size=n))


def build_records():
    rng = np.random.default_rng(20240901)
    recs = []  # (name, seq bytes)
    recs.append(("plain10k desc text", rand_seq(rng, 10000)))
    recs.append(("shorter_than_k", b"AC"))
    recs.append(("exactly_k3", b"ACG"))
    recs.append(("len4", b"ACGT"))
    recs.append(("len5", b"TTGCA"))
    recs.append(("len14", rand_seq(rng, 14)))
    recs.append(("len15", rand_seq(rng, 15)))
    recs.append(("len16", rand_seq(rng, 16)))
    recs.append(("empty", b""))
    s = bytearray(rand_seq(rng, 700))
    s[100:130] = b"N" * 30          # long N run
    s[300] = ord("N")               # isolated N
    s[316] = ord("N")               # two Ns 16 apart: exactly one valid 15-mer between
    s[340] = ord("N")
    s[355] = ord("N")               # 14 valid between: none
    recs.append(("n_runs", bytes(s)))
    recs.append(("lowercase", rand_seq(rng, 300).lower()))
    s = bytearray(rand_seq(rng, 400))
    s[50:80] = bytes(s[50:80]).lower()
    s[200:204] = b"RYKM"            # IUPAC
    recs.append(("mixed_case_iupac", bytes(s)))
    recs.append(("homopolymer_A", b"A" * 1200))
    recs.append(("homopolymer_G", b"G" * 257))
    recs.append(("dinuc", b"AT" * 300))
    recs.append(("palin4", b"ACGT" * 100))
    recs.append(("len63", rand_seq(rng, 63)))
    recs.append(("len64", rand_seq(rng, 64)))
    recs.append(("len65", rand_seq(rng, 65)))
    recs.append(("len4095", rand_seq(rng, 4095)))
    recs.append(("len4097", rand_seq(rng, 4097)))
    long_i = len(recs)
    recs.append(("len33000", rand_seq(rng, 33000)))
    # duplicated 100-base reads so 15-mer counts span every coverage branch
    bases = {}
    for c in (1, 2, 4, 5, 6, 7, 8, 12, 40, 41, 100):
        base = rand_seq(rng, 100)
        bases[c] = base
        for j in range(c):
            recs.append((f"dup{c}_{j}", base))
    # chimeric reads: pieces of differently-abundant families -> mixed histograms
    fams = sorted(bases)
    for j in range(12):
        parts = []
        for _ in range(int(rng.integers(3, 9))):
            b = bases[fams[int(rng.integers(0, len(fams)))]]
            ln = int(rng.integers(20, 70))
            st = int(rng.integers(0, 100 - ln + 1))
            parts.append(b[st:st + ln])
            if rng.random() < 0.3:
                parts.append(rand_seq(rng, int(rng.integers(5, 40))))
        recs.append((f"chimera{j}", b"".join(parts)))
    # one abundant 15-mer inside the 33 kb read: its bin fraction is < 1e-4 -> zeroed
    s = bytearray(recs[long_i][1])
    s[20000:20015] = bases[100][40:55]
    recs[long_i] = ("len33000", bytes(s))
    # reverse-complement pair: counts add across strands
    fw = rand_seq(rng, 200)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    recs.append(("fw", fw))
    recs.append(("rc", fw.translate(comp)[::-1]))
    return recs


def write_fasta(path, recs, width=70, eol=b"\n"):
    with open(path, "wb") as f:
        for i, (name, seq) in enumerate(recs):
            f.write(b">" + name.encode() + eol)
            w = width if i % 3 else 10 ** 9  # every third record single-line
            for j in range(0, len(seq), w):
                f.write(seq[j:j + w] + eol)
            if len(seq) == 0 and i % 2:
                f.write(eol)  # an empty line for some empty records


def write_fastq(path, recs):
    with open(path, "wb") as f:
        for name, seq in recs:
            f.write(b"@" + name.encode() + b"\n" + seq + b"\n+\n" + b"I" * len(seq) + b"\n")


def write_weird(path):
    """Header/format corner cases of the kseq-style reader."""
    with open(path, "wb") as f:
        f.write(b"junk before first header\n")
        f.write(b">r1 a comment > with gt\nACGTACGTACGTACGTACGT\n\n\nACGTTTGACCA\n")
        f.write(b">r2\tTabbed\nACGT@ACGT>ACGT+ACGTACGTACGTAAAC\n")   # @ > + inside a line are data
        f.write(b">\nGGGGGGGGGGGGGGGGGGGGGG\n")                          # empty name
        f.write(b">r4\n\r\nACGTACGTACGTACGTAC\r\nGT\r\n")                # CRLF + blank CRLF line first
        f.write(b">r5\nACGTAC GTACGTAC\tGTACGTAAC\n")                     # blanks inside sequence are data
        f.write(b">r6\n>r7\nTTTTTTTTTTTTTTTTTTTTTTTTTTACG\n")             # r6 empty
        f.write(b">r8 last without newline\nCCCCCCCCCCCCCCCCCCCCACGTAC")


def run(cmd):
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)


def sha(b):
    return hashlib.sha256(b).hexdigest()


def gz_write(path, data):
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(data)


def sparse_of_table(path):
    with open(path, "rb") as f:
        header = int(np.frombuffer(f.read(8), dtype="<u8")[0])
    t = np.memmap(path, dtype="<u4", mode="r", offset=8)
    idx_parts, cnt_parts = [], []
    step = 1 << 26
    for s in range(0, t.shape[0], step):
        blk = np.asarray(t[s:s + step])
        nz = np.flatnonzero(blk)
        if nz.size:
            idx_parts.append((nz + s).astype(np.uint32))
            cnt_parts.append(blk[nz].astype(np.uint32))
    idx = np.concatenate(idx_parts) if idx_parts else np.zeros(0, np.uint32)
    cnt = np.concatenate(cnt_parts) if cnt_parts else np.zeros(0, np.uint32)
    return header, int(t.shape[0]), idx, cnt


def main():
    if not os.path.exists(os.path.join(REF, "count-kmers")):
        sys.exit("oracle/_ref missing: run `make -C oracle` where /root/reference exists")
    recs = build_records()
    edge = os.path.join(HERE, "edge.fasta")
    write_fasta(edge, recs)
    write_fasta(os.path.join(HERE, "edge_crlf.fasta"), recs, eol=b"\r\n")
    write_fastq(os.path.join(HERE, "edge.fastq"), recs)
    with open(edge, "rb") as f:
        gz_write(os.path.join(HERE, "edge.fa.gz"), f.read())
    weird = os.path.join(HERE, "weird.fasta")
    write_weird(weird)

    meta = {"n_records_edge": len(recs), "outputs": {}}
    scratch = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=scratch) as tmp:
        def ref_com(inp, k):
            out = os.path.join(tmp, "com")
            run([os.path.join(REF, "count-kmers"), inp, out, str(k), "4"])
            return open(out, "rb").read()

        def ref_table(inp):
            out = os.path.join(tmp, "k15")
            run([os.path.join(REF, "count-15mers"), inp, out, "4"])
            return out

        def ref_cov(table, inp, bs, bc):
            out = os.path.join(tmp, "cov")
            run([os.path.join(REF, "search-15mers"), table, inp, out, str(bs), str(bc), "4"])
            return open(out, "rb").read()

        for k in (3, 4, 5):
            txt = ref_com(edge, k)
            gz_write(os.path.join(HERE, f"com_profs_k{k}.txt.gz"), txt)
            meta["outputs"][f"com_profs_k{k}"] = sha(txt)
        # input-format invariance, pinned on the reference itself
        for other in ("edge_crlf.fasta", "edge.fastq", "edge.fa.gz"):
            assert ref_com(os.path.join(HERE, other), 3) == ref_com(edge, 3), other

        table = ref_table(edge)
        header, size, idx, cnt = sparse_of_table(table)
        meta["table_header_word"] = header
        meta["table_entries"] = size
        meta["table_file_bytes"] = os.path.getsize(table)
        meta["table_sum"] = int(cnt.astype(np.uint64).sum())
        np.savez_compressed(os.path.join(HERE, "k15_sparse.npz"), idx=idx, cnt=cnt)
        for bs, bc in ((10, 32), (32, 10), (4, 10)):
            txt = ref_cov(table, edge, bs, bc)
            gz_write(os.path.join(HERE, f"cov_profs_bs{bs}_bc{bc}.txt.gz"), txt)
            meta["outputs"][f"cov_profs_bs{bs}_bc{bc}"] = sha(txt)
        os.remove(table)

        txt = ref_com(weird, 3)
        gz_write(os.path.join(HERE, "weird_com_k3.txt.gz"), txt)
        meta["outputs"]["weird_com_k3"] = sha(txt)
        table = ref_table(weird)
        _, _, idx, cnt = sparse_of_table(table)
        np.savez_compressed(os.path.join(HERE, "weird_k15_sparse.npz"), idx=idx, cnt=cnt)
        txt = ref_cov(table, weird, 4, 10)
        gz_write(os.path.join(HERE, "weird_cov_bs4_bc10.txt.gz"), txt)
        meta["outputs"]["weird_cov_bs4_bc10"] = sha(txt)
        os.remove(table)

    with open(os.path.join(HERE, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("golden fixtures written:", json.dumps(meta)[:300], "...")


if __name__ == "__main__":
    main()
