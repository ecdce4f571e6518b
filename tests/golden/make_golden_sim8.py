#!/usr/bin/env python3
"""Reference-scored accuracy gate on the 8-genome stand-in (tests/helpers.synth_sim8).

Build container only.  Drives the REFERENCE's own pipeline code (imported from /root/reference,
its three os.system shims pointed at the reference binaries in oracle/_ref) on the data set
SURVEY.md 8(d) describes: 8 genomes of 1-5 Mbp, abundances 5x-60x, 10 kb reads with ~10 % noise
(~40 k reads).  Flags are the README test run's (-k 3 -bc 10 --ae-dims 4 --ae-epochs 200 -bit 0)
with the histogram width and the minimum bin size scaled to the data (-bs 2 -mbs 500: the 15-mer
counts of a 5x-60x set with 10 % noise span 1..25, README.md:73 uses 32 for a ~100x set of 432 k
reads and -mbs 5000).  The reference is unseeded, so the harness seeds random/numpy/torch.

    make_golden_sim8.py run [seeds...]      whole reference pipeline per seed -> scores, and per seed
                                            the reference's latent.npy and, computed again from it under
                                            random.seed(seed), the reference's clusters and bins
                                            (tests/golden/sim8_ref_s{seed}.npz) for the stage-isolated test
    make_golden_sim8.py score DIR           DIR/latent_s{seed}.npy trained by THIS build on the GPU box
                                            (scripts/sim8_latents.py) -> the REFERENCE's perform_binning on
                                            them, scores next to the reference-trained ones

Writes tests/golden/e2e_reference_8g.json (numbers only) and sim8_ref_s*.npz (data only).
"""
import json
import os
import shutil
import subprocess
import sys
import time
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
import random

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)
from helpers import binning_scores, latent_pair_stats, synth_sim8, write_fasta  # noqa: E402
from make_golden_py import _parse  # noqa: E402

REFBIN = os.path.join(ROOT, "oracle", "_ref")
# SIM8_DATASET=blocks: the block-mixture data of round 1's Sim-8-scale run (helpers.synth_block_mixture,
# 40 k reads x 5 kb, README flags -bs 32 -mbs scaled) -> e2e_reference_blocks.json, scores only
BLOCKS = os.environ.get("SIM8_DATASET", "") == "blocks"
# SIM8_READS=432331: the same stand-in with genome lengths scaled to the read count of the real Sim-8 set
# (scripts/e2e_pipeline_scale.py runs exactly this on the GPU; -mbs 5000 as README.md:73) -> scores only
BIG = int(os.environ.get("SIM8_READS", "0"))
# SIM8_DATASET=c1: BASELINE config C1 on its OWN flags at its OWN size (README.md:73: -k 3 -bc 10 -bs 32 --ae-dims 4
# --ae-epochs 200 -bit 0 -mbs 5000; 432,333 reads): helpers.synth_sim8_c1, eight genomes of 100-600 kbp at
# 550x-3,100x -> e2e_reference_c1.json (scores, bins, wall time per seed)
C1 = os.environ.get("SIM8_DATASET", "") in ("c1", "c1hard")
# SIM8_DATASET=c1hard: the same flags and size on helpers.synth_sim8_c1_hard -- GC contents 2 % apart in pairs and two
# strains of one genome at different abundance -> e2e_reference_c1_hard.json: the set the reference itself strains on
C1HARD = os.environ.get("SIM8_DATASET", "") == "c1hard"
WORK = os.environ.get("SIM8_WORK", "/dev/shm/sim8_c1hard" if C1HARD else "/dev/shm/sim8_c1" if C1 else "/dev/shm/sim8_blocks" if BLOCKS else
                      "/dev/shm/sim8_big" if BIG else "/dev/shm/sim8_ref")
BS, BC, MBS, K, DIMS, EPOCHS = (32, 10, 5000, 3, 4, 200) if C1 else (32, 10, 100, 3, 4, 200) if BLOCKS else \
    (2, 10, 5000 if BIG else 500, 3, 4, 200)
# SIM8_JSON=path: write the run records there instead (several streams side by side, merged by `merge`)
JSON = os.environ.get("SIM8_JSON") or os.path.join(HERE, "e2e_reference_c1_hard.json" if C1HARD else "e2e_reference_c1.json" if C1 else "e2e_reference_blocks.json" if BLOCKS else
                    "e2e_reference_8g_big.json" if BIG else "e2e_reference_8g.json")


def import_reference():
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
    from mbcclr_utils import cluster_utils
    from mbcclr_utils import pipelines as P

    def sh(*cmd):
        subprocess.run(list(map(str, cmd)), check=True, stdout=subprocess.DEVNULL)

    P.run_kmers = lambda reads, out, k, t: sh(f"{REFBIN}/count-kmers", reads, f"{out}/profiles/com_profs", k, t)
    P.run_15mer_counts = lambda reads, out, t: sh(f"{REFBIN}/count-15mers", reads, f"{out}/profiles/15mers-counts", t)
    P.run_15mer_vecs = lambda reads, out, bs, bc, t: sh(f"{REFBIN}/search-15mers", f"{out}/profiles/15mers-counts",
                                                        reads, f"{out}/profiles/cov_profs", bs, bc, t)
    return P, cluster_utils


# where the latents of every run are kept (one directory for all streams of a data set)
LATENTS = os.environ.get("SIM8_LATENTS", WORK)


def pair_stats(latent, labels):
    """helpers.latent_pair_stats of the pairs that exist in this data set, rounded for the JSON."""
    from helpers import C1H_PAIRS, C1_PAIRS
    pairs = C1H_PAIRS if C1HARD else C1_PAIRS
    return {k: {q: round(x, 5) for q, x in v.items()} for k, v in latent_pair_stats(latent, labels, pairs).items()}


def seed_all(s):
    random.seed(s)
    np.random.seed(s)
    torch.manual_seed(s)


def dataset():
    os.makedirs(WORK, exist_ok=True)
    fa = os.path.join(WORK, "reads.fasta")
    lab = os.path.join(WORK, "labels.npy")
    if not (os.path.exists(fa) and os.path.exists(lab)):
        if C1HARD:
            from helpers import synth_sim8_c1_hard
            reads, labels = synth_sim8_c1_hard()
        elif C1:
            from helpers import synth_sim8_c1
            reads, labels = synth_sim8_c1()
        elif BLOCKS:
            from helpers import synth_block_mixture
            reads, labels = synth_block_mixture(40_000, glen=139_000)   # the coverage of the 432 k-read run (genomes 10.8x shorter)
        elif BIG:
            reads, labels = synth_sim8(scale=BIG / 40350.0)
        else:
            reads, labels = synth_sim8()
        write_fasta(fa, reads)
        np.save(lab, labels)
    return fa, np.load(lab)


def load_json():
    if os.path.exists(JSON):
        return json.load(open(JSON))
    return {}


def save_json(meta):
    runs = meta.get("runs", [])
    if runs:
        f1 = [r["f1"] for r in runs]
        meta["f1_mean"] = float(np.mean(f1))
        meta["f1_std"] = float(np.std(f1, ddof=1)) if len(f1) > 1 else 0.0
        meta["bins_median"] = float(np.median([r["bins"] for r in runs]))
    with open(JSON, "w") as f:
        json.dump(meta, f, indent=1)


def score(out, labels):
    bins = [int(x) for x in open(f"{out}/bins.txt").read().split()]
    p, r, f1, nb = binning_scores(bins, labels)
    from helpers import merged_genomes
    return {"precision": p, "recall": r, "f1": f1, "bins": nb, "merged": merged_genomes(bins, labels)}, np.array(bins)


def recluster(cluster_utils, out, seed, reads_path):
    """The reference's clustering stage alone on {out}/latent.npy under random.seed(seed)."""
    seed_all(seed)
    cluster_utils.perform_binning(out, 0, MBS, False, reads_path)


def run(seeds):
    P, cluster_utils = import_reference()
    fa, labels = dataset()
    out = os.path.join(WORK, "out")
    os.makedirs(os.path.join(out, "profiles"), exist_ok=True)
    args = types.SimpleNamespace(reads_path=fa, threads=8, bin_size=BS, bin_count=BC, k_size=K,
                                 ae_epochs=EPOCHS, ae_dims=DIMS, ae_hidden="128,128", separate=False,
                                 cuda=False, resume=True, min_bin_size=MBS, bin_iterations=0, output=out)
    meta = load_json()
    meta.update({"dataset": "helpers.synth_sim8_c1_hard()" if C1HARD else "helpers.synth_sim8_c1()" if C1 else
                 "helpers.synth_block_mixture(40000, glen=139000)" if BLOCKS else
                 f"helpers.synth_sim8(scale={BIG}/40350)" if BIG else "helpers.synth_sim8() defaults",
                 "n_reads": int(len(labels)),
                 "flags": f"-k {K} -bc {BC} -bs {BS} --ae-dims {DIMS} --ae-epochs {EPOCHS} -bit 0 -mbs {MBS}"})
    runs = {r["seed"]: r for r in meta.get("runs", [])}
    iso = {r["seed"]: r for r in meta.get("reference_latents_reclustered", [])}
    for seed in seeds:
        t0 = time.time()
        seed_all(seed)
        if os.path.exists(f"{out}/checkpoints"):  # force the VAE stage to run again, keep the profile stages
            ck = P.Checkpointer(f"{out}/checkpoints", True)
            ck.completed.pop("4_1", None)
            ck._save()
        P.run_reads_binning(args)
        res, _ = score(out, labels)
        res.update(seed=seed, wall_s=round(time.time() - t0, 1))
        runs[seed] = res
        print("reference e2e", res, flush=True)
        latent = np.load(f"{out}/latent.npy")
        np.save(os.path.join(LATENTS, f"ref_latent_s{seed}.npy"), latent)
        if C1:
            res["pairs"] = pair_stats(latent, labels)
        recluster(cluster_utils, out, seed, fa)
        res2, bins = score(out, labels)
        res2.update(seed=seed)
        iso[seed] = res2
        print("reference latents, clustering alone under random.seed", res2, flush=True)
        if not BLOCKS and not BIG and not C1:
            np.savez_compressed(os.path.join(HERE, f"sim8_ref_s{seed}.npz"), latent=latent.astype(np.float32),
                                bins=bins.astype(np.int16), seed=seed, mbs=MBS)
        meta["runs"] = [runs[s] for s in sorted(runs)]
        meta["reference_latents_reclustered"] = [iso[s] for s in sorted(iso)]
        save_json(meta)


def run_build_vae(seeds):
    """Isolating experiment (round 6): THIS build's torch-module VAE (lrbinner_amd.ae_utils, the LRB_VAE_NATIVE=0 code
    path) trained on the CPU of the build container from the reference binaries' profile files, then the REFERENCE's own
    perform_binning on the latents.  Everything but the VAE code is the reference's; nothing runs on a GPU.  If the
    strain-pair merges this build shows on the MI355X vanish here, they come from CPU-vs-GPU of shared code; if they
    stay, from what lrbinner_amd/ae_utils.py does differently from mbcclr_utils/ae_utils.py."""
    P, cluster_utils = import_reference()
    fa, labels = dataset()
    out = os.path.join(WORK, "out")
    assert os.path.exists(f"{out}/profiles/com_profs.npy"), "run the reference pipeline first (profile stages)"
    sys.path.insert(0, ROOT)
    from lrbinner_amd import ae_utils as build_ae
    meta = load_json()
    meta.update({"dataset": "helpers.synth_sim8_c1_hard()" if C1HARD else "helpers.synth_sim8_c1()", "n_reads": int(len(labels)),
                 "vae": "lrbinner_amd.ae_utils torch modules, device cpu", "clustering": "mbcclr_utils.cluster_utils.perform_binning",
                 "flags": f"-k {K} -bc {BC} -bs {BS} --ae-dims {DIMS} --ae-epochs {EPOCHS} -bit 0 -mbs {MBS}"})
    runs = {r["seed"]: r for r in meta.get("runs", [])}
    for seed in seeds:
        t0 = time.time()
        seed_all(seed)
        build_ae.vae_encode(out, DIMS, [128, 128], EPOCHS, None, False)
        t1 = time.time()
        latent = np.load(f"{out}/latent.npy")
        np.save(os.path.join(LATENTS, f"buildcpu_latent_s{seed}.npy"), latent)
        recluster(cluster_utils, out, seed, fa)
        res, _ = score(out, labels)
        res.update(seed=seed, vae_s=round(t1 - t0, 1), wall_s=round(time.time() - t0, 1), pairs=pair_stats(latent, labels))
        runs[seed] = res
        print("build VAE on the CPU, reference clustering", {k: v for k, v in res.items() if k != "pairs"}, flush=True)
        meta["runs"] = [runs[s] for s in sorted(runs)]
        save_json(meta)


def merge(paths):
    """Fold the run records of side streams (SIM8_JSON files) into this data set's committed JSON."""
    os.environ.pop("SIM8_JSON", None)
    meta = load_json()
    runs = {r["seed"]: r for r in meta.get("runs", [])}
    iso = {r["seed"]: r for r in meta.get("reference_latents_reclustered", [])}
    for p in paths:
        side = json.load(open(p))
        runs.update({r["seed"]: r for r in side.get("runs", [])})
        iso.update({r["seed"]: r for r in side.get("reference_latents_reclustered", [])})
    meta["runs"] = [runs[s] for s in sorted(runs)]
    meta["reference_latents_reclustered"] = [iso[s] for s in sorted(iso)]
    save_json(meta)
    print(len(runs), "runs in", JSON)


def add_pairs():
    """pair statistics for recorded runs whose latents are still in SIM8_LATENTS (runs made before round 6)."""
    _, labels = dataset()
    meta = load_json()
    for r in meta.get("runs", []):
        p = os.path.join(LATENTS, f"ref_latent_s{r['seed']}.npy")
        if "pairs" not in r and os.path.exists(p):
            r["pairs"] = pair_stats(np.load(p), labels)
    save_json(meta)


def score_latents(d):
    """THIS build's latents through the REFERENCE's clustering stage."""
    P, cluster_utils = import_reference()
    fa, labels = dataset()
    out = os.path.join(WORK, "out")
    assert os.path.exists(f"{out}/profiles/com_profs.npy"), "run the reference pipeline first"
    meta = load_json()
    res = []
    for name in sorted(os.listdir(d)):
        if not (name.startswith("latent_s") and name.endswith(".npy")):
            continue
        seed = int(name[len("latent_s"):-4])
        shutil.copy(os.path.join(d, name), f"{out}/latent.npy")
        recluster(cluster_utils, out, seed, fa)
        r, _ = score(out, labels)
        r.update(seed=seed)
        res.append(r)
        print("HIP-trained latents, reference clustering", r, flush=True)
    meta["hip_latents_reference_clustering"] = res
    if res:
        meta["hip_latents_reference_clustering_f1_mean"] = float(np.mean([r["f1"] for r in res]))
    save_json(meta)


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "run":
        run([int(s) for s in sys.argv[2:]] or [1, 2, 3])
    elif len(sys.argv) >= 3 and sys.argv[1] == "buildvae":
        run_build_vae([int(s) for s in sys.argv[2:]])
    elif len(sys.argv) >= 3 and sys.argv[1] == "merge":
        merge(sys.argv[2:])
    elif len(sys.argv) == 2 and sys.argv[1] == "addpairs":
        add_pairs()
    elif len(sys.argv) == 3 and sys.argv[1] == "score":
        score_latents(sys.argv[2])
    else:
        sys.exit(__doc__)
