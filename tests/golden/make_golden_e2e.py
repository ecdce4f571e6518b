#!/usr/bin/env python3
"""End-to-end reference run on the deterministic synthetic metagenome.

Build container only.  Drives the REFERENCE's own pipeline code
(mbcclr_utils.pipelines.run_reads_binning, imported from /root/reference) with its
three os.system runner shims pointed at the reference binaries built in oracle/_ref
(the reference looks for them in its own read-only tree), on the data set of
tests/helpers.synth_metagenome, the README test-run flags with the bin width scaled
to the data (-k 3 -bc 10 -bs 8 --ae-dims 4 --ae-epochs 200 -bit 0 -mbs 200).  The reference is
unseeded, so the harness seeds random/numpy/torch per repeat.  Writes the scores to
tests/golden/e2e_reference.json (numbers only).
"""
import json
import os
import subprocess
import sys
import tempfile
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
import random

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import binning_scores, synth_metagenome, write_fasta  # noqa: E402
from make_golden_py import _parse  # noqa: E402

REFBIN = os.path.join(ROOT, "oracle", "_ref")
MBS = 200


def main():
    bio = types.ModuleType("Bio")
    seqio = types.ModuleType("Bio.SeqIO")
    seqio.parse = _parse
    bio.SeqIO = seqio
    sys.modules["Bio"] = bio
    sys.modules["Bio.SeqIO"] = seqio
    sys.modules["metacoag_utils"] = types.ModuleType("metacoag_utils")
    sys.modules["metacoag_utils.marker_gene_utils"] = types.ModuleType("marker_gene_utils")
    sys.modules["metacoag_utils"].marker_gene_utils = sys.modules["metacoag_utils.marker_gene_utils"]
    sys.path.insert(0, "/root/reference")
    from mbcclr_utils import pipelines as P

    def sh(*cmd):
        subprocess.run(list(map(str, cmd)), check=True, stdout=subprocess.DEVNULL)

    P.run_kmers = lambda reads, out, k, t: sh(f"{REFBIN}/count-kmers", reads, f"{out}/profiles/com_profs", k, t)
    P.run_15mer_counts = lambda reads, out, t: sh(f"{REFBIN}/count-15mers", reads, f"{out}/profiles/15mers-counts", t)
    P.run_15mer_vecs = lambda reads, out, bs, bc, t: sh(f"{REFBIN}/search-15mers", f"{out}/profiles/15mers-counts",
                                                        reads, f"{out}/profiles/cov_profs", bs, bc, t)

    reads, labels = synth_metagenome()
    scratch = "/dev/shm" if os.path.isdir("/dev/shm") else None
    results = []
    with tempfile.TemporaryDirectory(dir=scratch) as tmp:
        fa = os.path.join(tmp, "reads.fasta")
        write_fasta(fa, reads)
        out = os.path.join(tmp, "out")
        os.makedirs(os.path.join(out, "profiles"))
        args = types.SimpleNamespace(reads_path=fa, threads=8, bin_size=8, bin_count=10, k_size=3,
                                     ae_epochs=200, ae_dims=4, ae_hidden="128,128", separate=False,
                                     cuda=False, resume=True, min_bin_size=MBS, bin_iterations=0,
                                     output=out)
        for rep, seed in enumerate((1, 2, 3)):
            random.seed(seed)
            np.random.seed(seed)
            torch.manual_seed(seed)
            if rep:  # force the VAE stage to run again, keep the profile stages
                ck = P.Checkpointer(f"{out}/checkpoints", True)
                ck.completed.pop("4_1", None)
                ck._save()
            P.run_reads_binning(args)
            bins = [int(x) for x in open(f"{out}/bins.txt").read().split()]
            p, r, f1, nb = binning_scores(bins, labels)
            results.append({"seed": seed, "precision": p, "recall": r, "f1": f1, "bins": nb})
            print(results[-1], flush=True)
            os.remove(f"{out}/profiles/15mers-counts") if rep == 2 else None
    meta = {"dataset": "helpers.synth_metagenome() defaults", "n_reads": len(reads),
            "flags": f"-k 3 -bc 10 -bs 8 --ae-dims 4 --ae-epochs 200 -bit 0 -mbs {MBS}",
            "runs": results, "f1_mean": float(np.mean([r["f1"] for r in results]))}
    with open(os.path.join(HERE, "e2e_reference.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print(json.dumps(meta)[:400])


if __name__ == "__main__":
    main()
