"""Stage orchestration -- the caller side of the hot path, same stage ids,
checkpoint parameters and file names as ``mbcclr_utils/pipelines.py`` so that
``--resume`` and every downstream consumer of ``{output}/`` behave the same.

reads   (pipelines.py:242-368): 1_1 composition, 1_2 15-mer table, 2_1 coverage,
        3_1 text -> npy, 4_1 VAE, then clustering (always re-run).
contigs (pipelines.py:13-240): lengths, fragments, table on the READS, profiles on the
        fragments, VAE, HDBSCAN over fragments + majority vote.  The marker-gene step
        needs FragGeneScan/HMMER and only feeds a loss term that the reference never
        activates; it is skipped with a log line.
"""
import logging
import os
import pickle
from collections import Counter, defaultdict

import numpy as np

from .runners_utils import (Checkpointer, contig_records, release_contigs, load_value_sidecar, run_15mer_counts,
                            run_15mer_vecs, run_kmers, split_contigs)
from . import ae_utils
from . import cluster_utils
from . import _npcache

logger = logging.getLogger('LRBinner')


def load_profile_text(path):
    """Text profile -> float64 [rows, cols]; same values as the reference's
    ``float(token)`` loop (pipelines.py:315-318) without the per-token Python.  When the
    runner that wrote the text left its value side-car, that is used instead of parsing."""
    side = load_value_sidecar(path)
    if side is not None:
        return side
    with open(path, "rb") as f:
        first = f.readline()
    cols = len(first.split())
    if cols == 0:
        return np.zeros((0, 0), dtype=np.float64)
    flat = np.fromfile(path, dtype=np.float64, sep=" ")
    return flat.reshape(-1, cols)


def _checkpoint(output, resume):
    path = f"{output}/checkpoints"
    if not resume:
        return Checkpointer(path)
    logger.info("Resuming the program from previous checkpoints")
    cp = Checkpointer(path, True)
    logger.debug(cp)
    return cp


def _stage(checkpoint, stage, params, start_msg, done_msg, skip_msg, fn):
    if checkpoint.should_run_step(stage, params):
        logger.info(start_msg)
        fn()
        checkpoint.log(stage, params)
        logger.info(done_msg)
    else:
        logger.info(skip_msg)


def _finish_table_file(output):
    """profiles/15mers-counts has been on its way to the disk since the counting stage (the
    library's writer thread); it has to be there before the run is over."""
    from .runners_utils import _guard, finish_table_files
    _guard("Counting 15-mers", lambda: finish_table_files(output))


def _profiles_to_npy(output):
    comp = load_profile_text(f"{output}/profiles/com_profs")
    cov = load_profile_text(f"{output}/profiles/cov_profs")
    _npcache.save(f"{output}/profiles/com_profs", comp)
    _npcache.save(f"{output}/profiles/cov_profs", cov)


def run_reads_binning(args):
    reads_path = args.reads_path
    threads = args.threads
    bin_size, bin_count, k_size = args.bin_size, args.bin_count, args.k_size
    epochs, dims = args.ae_epochs, args.ae_dims
    hidden = list(map(int, args.ae_hidden.split(",")))
    separate, cuda, resume = args.separate, args.cuda, args.resume
    min_cluster_size = max(args.min_bin_size, 1)
    iterations = max(args.bin_iterations, 0)
    output = args.output

    checkpoint = _checkpoint(output, resume)

    _stage(checkpoint, "1_1", [reads_path, k_size],
           "Counting k-mers", "Counting k-mers complete", "K-mer vectors already computed",
           lambda: run_kmers(reads_path, output, k_size, threads))
    _stage(checkpoint, "1_2", [reads_path],
           "Counting 15-mers", "Counting 15-mers complete", "15-mers already counted",
           lambda: run_15mer_counts(reads_path, output, threads, defer_table_file=True))
    _stage(checkpoint, "2_1", [reads_path, bin_size, bin_count],
           "Computing 15-mer profiles", "Computing 15-mer profiles complete",
           "Already computed 15-mer profiles complete",
           lambda: run_15mer_vecs(reads_path, output, bin_size, bin_count, threads))
    _stage(checkpoint, "3_1", ['numpy'],
           "Profiles saving as numpy arrays", "Profiles saving as numpy arrays complete",
           "Numpy arrays already computed", lambda: _profiles_to_npy(output))

    constraints = None

    def train():
        logger.info("VAE training information")
        logger.info(f"\tDimensions {dims}")
        logger.info(f"\tHidden Layers {hidden}")
        logger.info(f"\tEpochs {epochs}")
        ae_utils.vae_encode(output, dims, hidden, epochs, constraints, cuda)

    _stage(checkpoint, "4_1", [output, dims, hidden, epochs, constraints],
           "VAE training", "VAE training complete", "VAE already trained", train)

    _finish_table_file(output)
    cluster_utils.perform_binning(output, iterations, min_cluster_size, separate, reads_path)


def perform_contig_binning_HDBSCAN(output, fragment_parent, bincontigs, contigs_path, threads):
    """cluster_utils.py:483-537: HDBSCAN(min_cluster_size=250) on the fragment latents,
    each contig takes the most common label of its clustered fragments; contigs with
    no clustered fragment are left out of bins.txt.  The reference calls the
    third-party ``hdbscan`` package (version un-pinned: parity unpinned); here the published
    HDBSCAN* algorithm runs natively -- core distances and the mutual-reachability spanning
    tree as HIP kernels, the tree steps in the library's host code (include/lrb_hip.h K6) --
    with that package's defaults (min_samples = min_cluster_size, excess of mass).  A latent
    file with fewer rows than min_samples cannot be clustered (the package raises there
    too); every fragment is then noise."""
    from . import device as lrb
    latent = _npcache.load(f"{output}/latent.npy")
    if len(latent) >= 250:
        labels = lrb.Context(0).hdbscan(latent, min_cluster_size=250)
    else:
        logger.warning("fewer fragments than min_cluster_size: no clusters")
        labels = np.full(len(latent), -1, np.int32)
    logger.info(f"HDBSCAN detected {len(set(labels.tolist()) - {-1})}")
    votes = defaultdict(list)
    for frag, lab in enumerate(labels):
        if lab != -1:
            votes[fragment_parent[frag]].append(int(lab))
    contig_bin = {c: Counter(v).most_common()[0][0] for c, v in votes.items()}
    with open(f"{output}/binning_result.pkl", "wb+") as f:
        pickle.dump(contig_bin, f)
    bin_files = {}
    if bincontigs:
        os.makedirs(f"{output}/binned_contigs", exist_ok=True)
    with open(f"{output}/bins.txt", "w+") as out:
        for cid, seq in contig_records(contigs_path, want_seqs=bool(bincontigs)):
            if cid not in contig_bin:
                continue
            b = contig_bin[cid]
            out.write(f"{cid}\t{b}\n")
            if bincontigs:
                if b not in bin_files:
                    bin_files[b] = open(f"{output}/binned_contigs/Bin-{b}.fasta", "wb")
                bin_files[b].write(b">%b\n%b\n" % (cid.encode(), seq))
    for f in bin_files.values():
        f.close()
    release_contigs(contigs_path)


def run_contig_binning(args):
    reads_path, contigs = args.reads_path, args.contigs
    threads = args.threads
    bin_size, bin_count, k_size = args.bin_size, args.bin_count, args.k_size
    epochs, dims = args.ae_epochs, args.ae_dims
    hidden = list(map(int, args.ae_hidden.split(",")))
    separate, cuda, resume, output = args.separate, args.cuda, args.resume, args.output
    for sub in ("profiles", "fragments"):
        os.makedirs(f"{output}/{sub}", exist_ok=True)

    checkpoint = _checkpoint(output, resume)
    if checkpoint.should_run_step("1_1", ['contigs_binning']):
        checkpoint.log("1_1", ['contigs_binning'])

    def lengths():
        contig_length = {cid: len(seq) for cid, seq in contig_records(contigs)}
        with open(f"{output}/profiles/contig_lengths.pkl", "wb+") as f:
            pickle.dump(contig_length, f)

    _stage(checkpoint, "2_1", [contigs], "Computing contig lengths",
           "Computing contig lengths complete", "Contig lengths already computed", lengths)
    logger.info("Marker-gene constraints skipped (FragGeneScan/HMMER not part of this build; "
                "the reference's constraint loss is inactive)")

    state = {}

    def fragments():
        groups, parent = split_contigs(contigs, output)
        with open(f"{output}/fragments/fragment_parent.pkl", "wb+") as f:
            pickle.dump((dict(groups), parent), f)

    _stage(checkpoint, "2_3", [contigs, 'fragments'], "Splitting contigs", "Splitting contigs complete",
           "Contigs already split", fragments)
    with open(f"{output}/fragments/fragment_parent.pkl", "rb") as f:
        state["groups"], state["parent"] = pickle.load(f)
    frags = f"{output}/fragments/contigs.fasta"

    _stage(checkpoint, "3_1", [reads_path], "Counting 15-mers", "Counting 15-mers complete",
           "15-mers already counted", lambda: run_15mer_counts(reads_path, output, threads, defer_table_file=True))
    _stage(checkpoint, "3_2", [contigs, k_size], "Counting k-mers", "Counting k-mers complete",
           "K-mer vectors already computed", lambda: run_kmers(frags, output, k_size, threads))
    _stage(checkpoint, "3_3", [contigs, bin_size, bin_count], "Computing 15-mer profiles",
           "Computing 15-mer profiles complete", "Already computed 15-mer profiles complete",
           lambda: run_15mer_vecs(frags, output, bin_size, bin_count, threads))
    _stage(checkpoint, "4_1", ['numpy'], "Profiles saving as numpy arrays",
           "Profiles saving as numpy arrays complete", "Numpy arrays already computed",
           lambda: _profiles_to_npy(output))
    _stage(checkpoint, "5_1", [output, dims, hidden, epochs], "VAE training", "VAE training complete",
           "VAE already trained",
           lambda: ae_utils.vae_encode(output, dims, hidden, epochs, None, cuda))
    _finish_table_file(output)
    perform_contig_binning_HDBSCAN(output, state["parent"], separate, contigs, threads)
