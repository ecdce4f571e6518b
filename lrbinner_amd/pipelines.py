"""Stage orchestration -- the caller side of the hot path, same stage ids,
checkpoint parameters and file names as ``mbcclr_utils/pipelines.py`` so that
``--resume`` and every downstream consumer of ``{output}/`` behave the same.

reads   (pipelines.py:242-368): 1_1 composition, 1_2 15-mer table, 2_1 coverage,
        3_1 text -> npy, 4_1 VAE, then clustering (always re-run).
contigs (pipelines.py:13-240): 2_1 lengths (+ id <-> index maps), 2_3 fragments (+ the pair
        lists), 2_4 table on the READS, 3_1 / 4_1 profiles on the fragments, 5_1 npy, 6_1
        VAE, then HDBSCAN over fragments + majority vote (always re-run).  Stage 2_2, the
        marker-gene scan, needs FragGeneScan/HMMER and only feeds a loss term that is zero
        without marker hits; it is skipped with a log line and NOT logged as done, its
        product ``profiles/marker_contigs.pkl`` is written as an empty dict, so the pair
        lists are empty as they are in a reference run without marker hits.
"""
import logging
import os
import pickle
import shutil
from collections import Counter, defaultdict

import numpy as np

from .runners_utils import (Checkpointer, contig_lengths, contig_records, release_contigs, load_value_sidecar, run_15mer_counts,
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


def _stage(checkpoint, stage, params, start_msg, done_msg, skip_msg, fn, artifact=None):
    """One checkpointed stage (the if/else blocks of pipelines.py:270-366).  ``artifact``: a file (or
    several) the stage leaves behind LATER than its checkpoint -- the 15-mer table file is written by the
    library's writer thread while the following stages run, the two profile .npy files by _npcache's, and each
    appears under its name only when complete.  A run that died in between has the stage logged and no file: the stage then runs
    again on --resume, without touching the checkpoints of the stages after it (they consumed the
    same table from HBM)."""
    logged = not checkpoint.should_run_step(stage, params)
    artifacts = [] if artifact is None else [artifact] if isinstance(artifact, str) else list(artifact)
    if logged and all(os.path.exists(a) for a in artifacts):
        logger.info(skip_msg)
        return
    logger.info(start_msg)
    fn()
    if not logged:
        checkpoint.log(stage, params)
    logger.info(done_msg)


def _finish_table_file(output):
    """profiles/15mers-counts has been on its way to the disk since the counting stage (the
    library's writer thread); it has to be there before the run is over."""
    from .runners_utils import _guard, finish_table_files
    _guard("Counting 15-mers", lambda: finish_table_files(output))


_early = {}  # abs text path -> (thread, box): a profile text already on its way to float64 (_convert_early)


def _convert_early(checkpoint, stage, output, name):
    """Start turning profiles/<name> into its float64 array NOW, on a thread, although the stage that wants it
    (``stage``: 3_1 for reads, 5_1 for contigs) comes two stages later: com_profs is complete once the k-mer stage is
    through, the 15-mer stages that follow keep the GPU busy and the host idle, and the conversion is 0.8 of the 0.97 s
    stage 3_1 takes at C3 size (5 M x 136 doubles).  Only the array is made here -- the .npy file, the log lines and
    the checkpoint stay with the stage itself (_profiles_to_npy picks the array up), and only from the value
    side-car: parsing text holds the GIL and would slow the stages it is meant to hide behind."""
    path = os.path.abspath(f"{output}/profiles/{name}")
    due = checkpoint.should_run_step(stage, ['numpy']) or not all(os.path.exists(a) for a in _npy_artifacts(output))
    if not due or path in _early or not os.path.exists(path + ".q6.json"):
        return
    import threading
    box = {}

    def work():
        try:
            from .runners_utils import load_value_sidecar
            box["arr"] = load_value_sidecar(path)
        except BaseException:   # the stage does it again in line, and reports what fails there
            box["arr"] = None

    th = threading.Thread(target=work, daemon=True)
    _early[path] = (th, box)
    th.start()


def _profiles_to_npy(output):
    """pipelines.py:315-321.  The two arrays are made side by side (two threads: reading the side-car and the float64
    conversion release the GIL) and handed to the next stages in memory; the .npy files themselves (5.4 GB + 1.3 GB
    at C3 size, a third of the profile stages' wall time when written in line) go to the disk on writer threads
    while the VAE trains -- _finish_npy_files waits for them before the run ends."""
    from concurrent.futures import ThreadPoolExecutor

    def one(name):
        path = f"{output}/profiles/{name}"
        arr = None
        job = _early.pop(os.path.abspath(path), None)
        if job is not None:
            job[0].join()
            arr = job[1].get("arr")
        _npcache.save_async(path, arr if arr is not None else load_profile_text(path))

    with ThreadPoolExecutor(2) as pool:
        for f in [pool.submit(one, "com_profs"), pool.submit(one, "cov_profs")]:
            f.result()


def _npy_artifacts(output):
    return [f"{output}/profiles/com_profs.npy", f"{output}/profiles/cov_profs.npy"]


def _finish_npy_files():
    """The profile .npy files have been on their way to the disk since stage 3_1 / 5_1."""
    from .runners_utils import check_proc
    try:
        _npcache.finish()
    except OSError as e:
        logger.error(str(e))
        check_proc(1, "Profiles saving as numpy arrays")


def gpus_requested():
    """How the three profile stages are to run: ('launcher', world) when this process is one rank of a
    ``torch.distributed.run`` job, ('spawn', N) with LRB_GPUS=N > 1 in the environment (the stages then run as a
    child job of N ranks), else (None, 1).  SURVEY 8e: reads shard across the GPUs of one node, the 15-mer table
    is all-reduced once; everything after the profiles is single-GPU (the latent matrix is small)."""
    from . import dist as ld
    _, world, _ = ld.launcher_world()
    if world > 1:
        return "launcher", world
    try:
        n = int(os.environ.get("LRB_GPUS", "1"))
    except ValueError:
        n = 1
    return ("spawn", n) if n > 1 else (None, 1)


def _sharded_profile_stages(checkpoint, reads_path, output, k_size, bin_size, bin_count, threads):
    """Stages 1_1, 1_2 and 2_1 on several GPUs (lrbinner_amd.dist.profile_file_sharded: shards -> K1 + K2 ->
    fold -> all-reduce of the canonical half -> expand -> K3, every rank writing its rows at their final place), logged with the
    reference's stage ids and parameters (pipelines.py:269-302) so that --resume skips them like any other run.
    Taken when all three stages are due; a resume that needs only some of them runs those on one GPU.
    Returns True when the stages were handled here."""
    from . import dist as ld
    from .runners_utils import check_proc
    mode, world = gpus_requested()
    if mode is None:
        return False
    stages = [("1_1", [reads_path, k_size]), ("1_2", [reads_path]), ("2_1", [reads_path, bin_size, bin_count])]
    due = [checkpoint.should_run_step(s, p) for s, p in stages]
    due[1] = due[1] or not os.path.exists(f"{output}/profiles/15mers-counts")
    take = all(due)
    if mode == "launcher":
        # every rank is here (the others through run_profile_rank): rank 0's verdict is the common one
        local = ld.init_group()
        box = [take]
        ld._dist().broadcast_object_list(box, src=0)
        take = bool(box[0])
    if not take:
        if mode == "launcher":
            ld.close_group()
        if any(due):
            logger.info("Resuming part of the profile stages: they run on one GPU")
        return False
    logger.info(f"Profile stages on {world} GPUs: reads sharded, one all-reduce of the 15-mer table")
    logger.info("Counting k-mers")
    logger.info("Counting 15-mers")
    logger.info("Computing 15-mer profiles")
    if mode == "spawn":
        ret = ld.spawn_ranks(world, ["--reads", reads_path, "--output", output, "-k", k_size, "-bs", bin_size,
                                     "-bc", bin_count, "-t", threads])
        check_proc(ret, "Profiles (multi-GPU)")
    else:
        try:
            ld.profile_file_sharded(reads_path, output, k_size, bin_size, bin_count, threads, ld.HipCompute(local))
        finally:
            ld.close_group()
    for (stage, params), msg in zip(stages, ("Counting k-mers complete", "Counting 15-mers complete",
                                             "Computing 15-mer profiles complete")):
        checkpoint.log(stage, params)
        logger.info(msg)
    return True


def run_profile_rank(args):
    """A rank other than 0 of ``torch.distributed.run ... lrbinner.py reads``: its share of the three profile
    stages when rank 0 decides to run them sharded, nothing else (VAE and clustering are rank 0's)."""
    from . import dist as ld
    local = ld.init_group()
    box = [None]
    ld._dist().broadcast_object_list(box, src=0)
    try:
        if box[0]:
            ld.profile_file_sharded(args.reads_path, args.output, args.k_size, args.bin_size, args.bin_count,
                                    args.threads, ld.HipCompute(local))
    finally:
        ld.close_group()


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

    _sharded_profile_stages(checkpoint, reads_path, output, k_size, bin_size, bin_count, threads)
    _stage(checkpoint, "1_1", [reads_path, k_size],
           "Counting k-mers", "Counting k-mers complete", "K-mer vectors already computed",
           lambda: run_kmers(reads_path, output, k_size, threads))
    _convert_early(checkpoint, "3_1", output, "com_profs")
    _stage(checkpoint, "1_2", [reads_path],
           "Counting 15-mers", "Counting 15-mers complete", "15-mers already counted",
           lambda: run_15mer_counts(reads_path, output, threads, defer_table_file=True, coverage_bins=bin_count),
           artifact=f"{output}/profiles/15mers-counts")
    _stage(checkpoint, "2_1", [reads_path, bin_size, bin_count],
           "Computing 15-mer profiles", "Computing 15-mer profiles complete",
           "Already computed 15-mer profiles complete",
           lambda: run_15mer_vecs(reads_path, output, bin_size, bin_count, threads))
    _stage(checkpoint, "3_1", ['numpy'],
           "Profiles saving as numpy arrays", "Profiles saving as numpy arrays complete",
           "Numpy arrays already computed", lambda: _profiles_to_npy(output), artifact=_npy_artifacts(output))

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
    _finish_npy_files()
    cluster_utils.perform_binning(output, iterations, min_cluster_size, separate, reads_path)


def contig_votes(labels, fragment_parent):
    """contig id -> bin, in the reference's order (cluster_utils.py:496-515): clusters in order of
    first appearance of their label, their fragments in index order; a contig's candidates are
    listed in that order, and Counter.most_common breaks ties by first insertion -- so a 2-vs-2
    contig goes to the label that appeared first in the FILE, not first in the contig.  The dict's
    order is the row order of bins.txt."""
    lab = np.asarray(labels).astype(np.int64)
    n = len(lab)
    valid = np.flatnonzero(lab != -1)
    if len(valid) == 0:
        return {}
    # the same walk, vectorised (1.7 M fragments at C5 size): rank of a label = its position in first-appearance order
    uniq, first_idx, inv = np.unique(lab[valid], return_index=True, return_inverse=True)
    rank_of = np.empty(len(uniq), dtype=np.int64)
    rank_of[np.argsort(first_idx, kind="stable")] = np.arange(len(uniq))
    label_by_rank = np.empty(len(uniq), dtype=np.int64)
    label_by_rank[rank_of] = uniq
    rank = rank_of[inv]
    parents = list(map(fragment_parent.__getitem__, valid.tolist()))
    index_of = dict.fromkeys(parents)                 # contig ids in order of first appearance ...
    for i, cid in enumerate(index_of):
        index_of[cid] = i                             # ... numbered
    pidx = np.fromiter(map(index_of.__getitem__, parents), dtype=np.int64, count=len(parents))
    n_contigs, n_labels = len(index_of), len(uniq)
    # votes per (contig, label); the winner has the most, ties go to the label of the earliest cluster
    ukey, cnt = np.unique(pidx * n_labels + rank, return_counts=True)
    uc, ur = ukey // n_labels, ukey % n_labels
    o = np.lexsort((ur, -cnt, uc))
    uc, ur = uc[o], ur[o]
    lead = np.r_[True, uc[1:] != uc[:-1]]
    best = np.empty(n_contigs, dtype=np.int64)
    best[uc[lead]] = label_by_rank[ur[lead]]
    # a contig enters the dict when the walk first meets it: its smallest (cluster rank, fragment index)
    met = np.full(n_contigs, np.iinfo(np.int64).max, dtype=np.int64)
    np.minimum.at(met, pidx, rank * n + valid)
    ids = list(index_of)
    return {ids[c]: int(best[c]) for c in np.argsort(met, kind="stable").tolist()}


def perform_contig_binning_HDBSCAN(output, fragment_parent, bincontigs, contigs_path, threads):
    """cluster_utils.py:483-537: HDBSCAN(min_cluster_size=250) on the fragment latents,
    each contig takes the most common label of its clustered fragments (contig_votes); contigs
    with no clustered fragment are left out of bins.txt and, with --separate, go to
    binned_contigs/Bin-unbinned.fasta.  The reference calls the
    third-party ``hdbscan`` package (version un-pinned: parity unpinned); here the published
    HDBSCAN* algorithm runs natively -- core distances and the mutual-reachability spanning
    tree as HIP kernels, the tree steps in the library's host code (include/lrb_hip.h K6) --
    with that package's defaults (min_samples = min_cluster_size cut to n - 1, excess of mass; core
    distances to the min_samples-th OTHER point as the package's Boruvka paths take them --
    LRB_HDB_CORE=self for the other convention).  With fewer rows than
    min_cluster_size no cluster can form: every fragment is noise, as from the package."""
    from . import device as lrb
    latent = _npcache.load(f"{output}/latent.npy")
    if len(latent) >= 2:
        labels = lrb.Context(0).hdbscan(latent, min_cluster_size=250)
    else:
        labels = np.full(len(latent), -1, np.int32)
    logger.info(f"HDBSCAN detected {len(set(labels.tolist()) - {-1})}")
    contig_bin = contig_votes(labels, fragment_parent)
    with open(f"{output}/binning_result.pkl", "wb+") as f:
        pickle.dump(contig_bin, f)
    with open(f"{output}/bins.txt", "w+") as out:
        for cid, b in contig_bin.items():
            out.write(f"{cid}\t{b}\n")
    if bincontigs:
        logger.info("Separating contigs into bin files")
        if os.path.isdir(f"{output}/binned_contigs"):
            shutil.rmtree(f"{output}/binned_contigs")
        os.mkdir(f"{output}/binned_contigs")
        bin_files = {}
        for cid, seq in contig_records(contigs_path, want_seqs=True):
            b = contig_bin.get(cid, "unbinned")     # cluster_utils.py:511: defaultdict(lambda: 'unbinned')
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
        contig_length, id_idx, idx_id = {}, {}, {}
        for cid, length in zip(*contig_lengths(contigs)):
            contig_length[cid] = length
            idx_id[len(id_idx)] = cid
            id_idx[cid] = len(id_idx)
        for name, obj in (("contig_lengths", contig_length), ("contig_id_idx", id_idx), ("contig_idx_id", idx_id)):
            with open(f"{output}/profiles/{name}.pkl", "wb+") as f:
                pickle.dump(obj, f)

    _stage(checkpoint, "2_1", [contigs], "Computing contig lengths",
           "Computing contig lengths complete", "Loading contig lengths", lengths)
    # 2_2 (pipelines.py:66-85) is not logged: a reference run resuming this directory does its own scan
    logger.info("Marker-gene constraints skipped (FragGeneScan/HMMER not part of this build; "
                "without marker hits the reference's constraint loss is zero)")
    if not os.path.exists(f"{output}/profiles/marker_contigs.pkl"):
        with open(f"{output}/profiles/marker_contigs.pkl", "wb+") as f:
            pickle.dump({}, f)

    def fragments():
        groups, parent = split_contigs(contigs, output)
        for name, obj in (("must_link_pairs", []), ("must_not_link_pairs", []),
                          ("contig_groups", groups), ("fragment_parent", parent)):
            with open(f"{output}/profiles/{name}.pkl", "wb+") as f:
                pickle.dump(obj, f)

    _stage(checkpoint, "2_3", [contigs], "Splitting contigs", "Splitting contigs completed",
           "Contigs already split", fragments)
    with open(f"{output}/profiles/fragment_parent.pkl", "rb") as f:
        fragment_parent = pickle.load(f)
    frags = f"{output}/fragments/contigs.fasta"

    _stage(checkpoint, "2_4", [reads_path], "Counting 15-mers", "Counting 15-mers complete",
           "15-mer counting already performed",
           lambda: run_15mer_counts(reads_path, output, threads, defer_table_file=True),
           artifact=f"{output}/profiles/15mers-counts")
    _stage(checkpoint, "3_1", [frags, k_size], "Computing k-mer vectors", "Computing k-mer vectors complete",
           "K-mer vectors already computed", lambda: run_kmers(frags, output, k_size, threads))
    _convert_early(checkpoint, "5_1", output, "com_profs")
    _stage(checkpoint, "4_1", [frags, bin_size, bin_count], "Generating coverage vectors",
           "Generating coverage vectors complete", "Coverage vectors already computed",
           lambda: run_15mer_vecs(frags, output, bin_size, bin_count, threads))
    _stage(checkpoint, "5_1", ['numpy'], "Profiles saving as numpy arrays",
           "Profiles saving as numpy arrays complete", "Numpy arrays already computed",
           lambda: _profiles_to_npy(output), artifact=_npy_artifacts(output))

    def train():
        logger.info("VAE training information")
        logger.info(f"\tDimensions {dims}")
        logger.info(f"\tHidden Layers {hidden}")
        logger.info(f"\tEpochs {epochs}")
        logger.info(f"Contig split must link pairs   {0:10}")
        logger.info(f"Single copy marker genes pairs {0:10}")
        # the reference passes {'ml': [], 'mnl': [], 'size': N} here; with both lists empty its
        # two constraint terms are 0 (ae_utils.py:247-255), which is what None computes
        ae_utils.vae_encode(output, dims, hidden, epochs, None, cuda)

    _stage(checkpoint, "6_1", [output, dims, hidden, epochs, 0, 0], "VAE training", "VAE training complete",
           "VAE already trained", train)
    _finish_table_file(output)
    _finish_npy_files()
    perform_contig_binning_HDBSCAN(output, fragment_parent, separate, contigs, threads)
