"""Worker for tests/test_dist_gloo.py: runs lrbinner_amd.dist under gloo with a small
CPU stand-in for the GPU compute object (test infrastructure; the stand-in is a
7-mer miniature of the 15-mer table path so that two ranks fit in memory)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle as orc  # noqa: E402
from lrbinner_amd import dist as ld  # noqa: E402

KW = 7
ENTRIES = 4 ** KW


def windows(seq):
    """forward codes of the valid KW-mers of one read (ACGT-only windows)."""
    out, val, run = [], 0, 0
    for c in seq:
        if c not in b"ACGT":
            val, run = 0, 0
            continue
        val = ((val << 2) & (ENTRIES - 1)) + ((c >> 1) & 3)
        run = min(run + 1, KW)
        if run == KW:
            out.append(val)
    return np.array(out, dtype=np.int64)


class MiniCompute:
    def __init__(self, preload=0):
        self.preload = preload

    def new_table(self):
        t = torch.zeros(ENTRIES, dtype=torch.int32)
        if self.preload and dist.get_rank() == 0:
            t[:] = self.preload  # near the uint32 limit: the sum has to wrap
        return t

    def kmer_counts(self, seqs, offs, k):
        return orc.count_kmers(np.ascontiguousarray(seqs), np.ascontiguousarray(offs), k)[0]

    def k15_accumulate(self, seqs, offs, table):
        tv = table.numpy().view(np.uint32)
        for r in orc.reads_of(seqs, offs):
            np.add.at(tv, windows(r), np.uint32(1))

    def k15_mirror(self, table):
        tv = table.numpy().view(np.uint32)
        rc = np.array([orc.revcomp(x, KW) for x in range(ENTRIES)], dtype=np.int64)
        tv[:] = tv + tv[rc]

    # canonical half of the miniature: of x and rc(x) the one whose middle base (bits 7..6 of a
    # 7-mer) has high bit 0, numbered by dropping that bit -- lrb_k15_fold_half_dev in small
    def k15_fold_half(self, table):
        tv = table.numpy().view(np.uint32)
        x = np.arange(ENTRIES, dtype=np.int64)
        canon = x[(x >> 7) & 1 == 0]
        rc = np.array([orc.revcomp(int(v), KW) for v in canon], dtype=np.int64)
        assert (((rc >> 7) & 1) == 1).all()
        h = ((canon >> 8) << 7) | (canon & 0x7F)
        half = np.zeros(ENTRIES // 2, dtype=np.uint32)
        half[h] = tv[canon] + tv[rc]
        self.folds = getattr(self, "folds", 0) + 1
        return torch.from_numpy(half.view(np.int32))

    def k15_expand_half(self, half, table):
        tv = table.numpy().view(np.uint32)
        hv = half.numpy().view(np.uint32)
        x = np.arange(ENTRIES, dtype=np.int64)
        canon = x[(x >> 7) & 1 == 0]
        rc = np.array([orc.revcomp(int(v), KW) for v in canon], dtype=np.int64)
        h = ((canon >> 8) << 7) | (canon & 0x7F)
        tv[canon] = hv[h]
        tv[rc] = hv[h]

    def sync(self):
        pass

    def cov_hist(self, seqs, offs, table, bin_size, bins):
        tv = table.numpy().view(np.uint32)
        n = len(offs) - 1
        hist = np.zeros((n, bins), dtype=np.uint32)
        sums = np.zeros(n, dtype=np.uint32)
        for i, r in enumerate(orc.reads_of(seqs, offs)):
            w = windows(r)
            sums[i] = len(w)
            for c in tv[w]:
                hist[i, orc.cov_bin(int(c), bin_size, bins)] += 1
        return hist, sums


class MiniPacked:
    """Stand-in for a batch kept in HBM: holds copies of the host arrays."""

    def __init__(self, comp, seqs, offs):
        self.comp, self.seqs, self.offs = comp, np.array(seqs, copy=True), np.array(offs, copy=True)
        self.n, self.device_bytes = len(offs) - 1, int(offs[-1]) + 1

    def kmer_counts(self, k):
        return self.comp.kmer_counts(self.seqs, self.offs, k)

    def k15_accumulate(self, table):
        self.comp.k15_accumulate(self.seqs, self.offs, table)

    def cov_hist(self, table, bin_size, bins):
        return self.comp.cov_hist(self.seqs, self.offs, table, bin_size, bins)

    def free(self):
        self.seqs = self.offs = None


class PackingCompute(MiniCompute):
    """MiniCompute that can keep batches: budget in bytes of sequence."""

    def __init__(self, budget):
        super().__init__()
        self.budget, self.packed = budget, 0

    def resident_budget(self):
        return self.budget

    def pack(self, seqs, offs, k):
        self.packed += 1
        return MiniPacked(self, seqs, offs)


def main():
    mode, reads_path, out = sys.argv[1:4]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    if mode in ("array", "array_full"):
        if mode == "array_full":
            os.environ["LRB_ALLREDUCE"] = "full"    # the whole-table all-reduce, for A/B
        buf, offs = orc.fastx_read(reads_path)
        preload = 0xFFFFFFF0 - (1 << 32)  # as int32
        comp = MiniCompute(preload)
        lo, hi, counts, hist, sums = ld.profile_reads_sharded(buf, offs, 3, 4, 10, comp)
        # the half-table form is what runs when there is more than one rank (and only then)
        assert getattr(comp, "folds", 0) == (1 if world > 1 and mode == "array" else 0)
        allc = ld.gather_rows(counts)
        allh = ld.gather_rows(hist)
        alls = ld.gather_rows(sums)
        if rank == 0:
            np.savez(out, counts=allc, hist=allh, sums=alls, world=world)
    else:
        # "file": no residency (two parses); "file_keep": everything stays packed;
        # "file_spill": the budget runs out half way, the rest is parsed again in phase B
        comp = {"file": MiniCompute(), "file_keep": PackingCompute(1 << 40),
                "file_spill": PackingCompute(4000)}[mode]
        nb = ld.profile_file_sharded(reads_path, out, 3, 4, 10, 2, comp, batch_reads=37,
                                     write_table=False, chunk_bytes=700)
        if rank == 0:
            open(os.path.join(out, "nbatches"), "w").write(str(nb))
        if mode != "file":
            assert comp.packed > 0
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
