"""The accuracy gate at realistic scale (north_star: F1 within +-0.5 of the reference): the 8-genome
stand-in for Sim-8 (tests/helpers.synth_sim8: 8 genomes of 1-5 Mbp, 5x-60x, 10 kb reads with ~10 %
noise, 40,350 reads), scored as eval.py:37-45 does, against tests/golden/e2e_reference_8g.json -- the
REFERENCE's own pipeline run on the same reads in the build container (make_golden_sim8.py, >= 3
seeds; README.md:78-95 reports 98.12 / 8 bins on the real Sim-8).  Plus the stage-isolated checks
that keep run-to-run noise from hiding a defect in one stage."""
import json
import os
import random
import shutil
import subprocess
import sys

import numpy as np
import pytest

from helpers import ROOT, binning_scores, golden_path, synth_sim8, write_fasta

pytestmark = pytest.mark.gpu

FLAGS = ["-k", "3", "-bc", "10", "-bs", "2", "--ae-dims", "4", "--ae-epochs", "200", "-bit", "0", "-mbs", "500"]
SEEDS = (1, 2, 3, 4, 5)


@pytest.fixture(scope="module")
def sim8(tmp_path_factory):
    d = tmp_path_factory.mktemp("sim8")
    reads, labels = synth_sim8()
    fa = str(d / "reads.fasta")
    write_fasta(fa, reads)
    return fa, labels, d


@pytest.fixture(scope="module")
def runs(sim8):
    """lrbinner.py reads --cuda once per seed; the output directories are kept for the tests below."""
    fa, labels, d = sim8
    keep = os.path.join(ROOT, "gpurun_out", "sim8_latents")   # scratch that travels back to the build container
    try:
        os.makedirs(keep, exist_ok=True)
    except OSError:
        keep = None
    out = {}
    for seed in SEEDS:
        o = str(d / f"out{seed}")
        cmd = [sys.executable, os.path.join(ROOT, "lrbinner.py"), "reads", "-r", fa, "-o", o] + FLAGS + ["--cuda", "-t", "8"]
        subprocess.run(cmd, check=True, cwd=ROOT, env=dict(os.environ, LRB_SEED=str(seed)))
        bins = [int(x) for x in open(os.path.join(o, "bins.txt")).read().split()]
        p, r, f1, nb = binning_scores(bins, labels)
        out[seed] = {"dir": o, "precision": p, "recall": r, "f1": f1, "bins": nb}
        print("sim8 e2e seed", seed, out[seed])
        if keep:
            shutil.copy(os.path.join(o, "latent.npy"), os.path.join(keep, f"latent_s{seed}.npy"))  # -> make_golden_sim8.py score
        if os.path.exists(os.path.join(o, "profiles/15mers-counts")):
            os.remove(os.path.join(o, "profiles/15mers-counts"))
    if keep:
        with open(os.path.join(keep, "e2e_scores.json"), "w") as f:
            json.dump({str(s): {k: v for k, v in r.items() if k != "dir"} for s, r in out.items()}, f, indent=1)
    return out


def test_sim8_end_to_end_f1_and_bins_vs_reference(sim8, runs):
    """Five seeded runs against the reference's five: the MEDIAN F1 within +-0.5 of the reference's median
    (north_star's tolerance; the reference's own sigma is 0.01), the median number of bins equal, and at
    least three runs individually within +-0.5 of the reference's mean.  Medians, not means: a seeded run of
    this build repeats only statistically (float atomics in the VAE's batch statistics), and about one run in
    fifteen merges two of the eight genomes (7 bins, F1 96.8) -- as the reference does too (2 of its 5 runs at
    the closer GC spacing, tests/golden/e2e_reference_8g_close.json) -- which moves a five-run MEAN by 0.6:
    a gate on the mean would fail one time in three on a build that is right.  The mean is printed."""
    ref = json.load(open(golden_path("e2e_reference_8g.json")))
    assert ref["n_reads"] == len(sim8[1]) and ref["flags"] == " ".join(FLAGS)
    f1 = np.array([r["f1"] for r in runs.values()])
    ref_f1 = np.array([r["f1"] for r in ref["runs"]])
    print("sim8 F1 median", np.median(f1), "mean", f1.mean(), "| reference median", np.median(ref_f1), "mean",
          ref["f1_mean"], "+-", ref["f1_std"])
    assert abs(np.median(f1) - np.median(ref_f1)) <= 0.5
    assert np.median([r["bins"] for r in runs.values()]) == ref["bins_median"]
    assert int((np.abs(f1 - ref["f1_mean"]) <= 0.5).sum()) >= 3


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_sim8_reference_latents_through_this_clustering(sim8, runs, seed):
    """Stage isolation (i): the latent.npy the REFERENCE trained (tests/golden/sim8_ref_s{seed}.npz), put
    into an output directory holding THIS build's profiles of the same reads, through this build's
    cluster_points + perform_binning under random.seed(seed): the reference's own bins.txt for that
    latent and seed (same file) must come out -- same number of bins, > 99.5 % of the reads in the
    same bin (the two differ in float32 summation order of the distances only)."""
    from lrbinner_amd import cluster_utils
    fa, labels, d = sim8
    z = np.load(golden_path(f"sim8_ref_s{seed}.npz"))
    o = str(d / f"iso{seed}")
    shutil.copytree(runs[SEEDS[0]]["dir"], o)
    for f in ("bins.txt", "binning_result.pkl", "lengths.txt"):
        os.remove(os.path.join(o, f))
    np.save(os.path.join(o, "latent.npy"), z["latent"])
    random.seed(int(z["seed"]))
    cluster_utils.perform_binning(o, 0, int(z["mbs"]), False, fa)
    bins = np.array([int(x) for x in open(os.path.join(o, "bins.txt")).read().split()])
    want = z["bins"].astype(np.int64)
    assert len(set(bins.tolist())) == len(set(want.tolist()))
    agree = float((bins == want).mean())
    print("sim8 stage (i) seed", seed, "agreement", agree, binning_scores(bins, labels))
    assert agree > 0.995


# ---- BASELINE config C1 on its OWN flags at its OWN size -----------------------------------------
C1_FLAGS = ["-k", "3", "-bc", "10", "-bs", "32", "--ae-dims", "4", "--ae-epochs", "200", "-bit", "0", "-mbs", "5000"]


def test_c1_own_flags_own_size_vs_reference(tmp_path):
    """README.md:73's test run as it stands -- `-k 3 -bc 10 -bs 32 --ae-dims 4 --ae-epochs 200 -bit 0 -mbs 5000` --
    on a stand-in of Sim-8's own size: helpers.synth_sim8_c1, 432,333 reads of 10 kb from eight genomes at
    550x-3,100x, so that the 15-mer counts run into the hundreds and the README's histogram (10 bins of width 32)
    carries signal.  tests/golden/e2e_reference_c1.json holds the REFERENCE's own pipeline on the same reads (build
    container, one run per seed, ~48 min each).  Five seeded runs of this build: the median F1 within +-0.5 of the
    reference's median (north_star's tolerance), the median number of bins equal, at least three runs individually
    within +-0.5 of the reference's mean -- and the number of runs that end with fewer than eight bins is printed
    and recorded next to the reference's (the cluster search both share merges two neighbouring genomes in a
    fraction of its runs)."""
    from helpers import synth_sim8_c1
    ref = json.load(open(golden_path("e2e_reference_c1.json")))
    reads, labels = synth_sim8_c1()
    assert ref["n_reads"] == len(reads) == 432_333 and ref["flags"] == " ".join(C1_FLAGS)
    fa = str(tmp_path / "reads.fasta")
    write_fasta(fa, reads)
    del reads
    res = []
    for seed in SEEDS:
        o = str(tmp_path / f"out{seed}")
        cmd = [sys.executable, os.path.join(ROOT, "lrbinner.py"), "reads", "-r", fa, "-o", o] + C1_FLAGS + ["--cuda", "-t", "32"]
        t0 = __import__("time").time()
        subprocess.run(cmd, check=True, cwd=ROOT, env=dict(os.environ, LRB_SEED=str(seed)))
        wall = __import__("time").time() - t0
        bins = [int(x) for x in open(os.path.join(o, "bins.txt")).read().split()]
        p, r, f1, nb = binning_scores(bins, labels)
        res.append({"seed": seed, "precision": p, "recall": r, "f1": f1, "bins": nb, "wall_s": round(wall, 1)})
        print("C1 e2e", res[-1])
        shutil.rmtree(o)
    f1 = np.array([r["f1"] for r in res])
    ref_f1 = np.array([r["f1"] for r in ref["runs"]])
    few = sum(r["bins"] < 8 for r in res)
    ref_few = sum(r["bins"] < 8 for r in ref["runs"])
    print(f"C1: F1 median {np.median(f1):.3f} mean {f1.mean():.3f} | reference median {np.median(ref_f1):.3f} "
          f"({len(ref_f1)} runs) | runs with < 8 bins: {few} of {len(res)} (reference {ref_few} of {len(ref_f1)})")
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "c1_e2e_scores.json"), "w") as f:
            json.dump({"runs": res, "runs_below_8_bins": few, "reference_runs_below_8_bins": ref_few,
                       "reference_f1": ref_f1.tolist()}, f, indent=1)
    except OSError:
        pass
    assert abs(np.median(f1) - np.median(ref_f1)) <= 0.5
    assert np.median([r["bins"] for r in res]) == ref["bins_median"]
    assert int((np.abs(f1 - ref["f1_mean"]) <= 0.5).sum()) >= 3


# ---- an accuracy set on which the method STRAINS (round 4; statistics round 5) -------------------------------
from helpers import binom_sf as _binom_sf, hard_set_statistics  # noqa: E402


def test_c1_hard_strains_vs_reference(tmp_path):
    """C1's size and flags on helpers.synth_sim8_c1_hard (the eighth genome a 10 %-diverged STRAIN of the seventh at three
    times its abundance: 3-mer composition cannot tell them apart, the 15-mer coverage histogram has to), five runs of this
    build under LRB_VAE_DETERMINISTIC=1 (a seed's outcome is then a fixed fact of the build).  What five runs can hold:
      * every run ends in one of the four outcomes the reference's own latents show (all eight genomes; the strain pair in
        one bin, F1 97.2; genomes 5 and 7 in one bin, 92.3; both, 89.2) at exactly the F1 that outcome costs;
      * an 8-bin run lies within +-0.5 F1 of a reference 8-bin run;
      * at most q runs below eight bins, q = the 99 % quantile of Binomial(5, p), p = the upper 95 % bound of the rate at
        which the REFERENCE's latents merge a pair per search (34 of 281 searches under eleven seeds,
        profiles/r06_c1hard_ref_recluster_*.json) -- the reference's bound, not this build's.
    This gate catches a dead coverage path (the pair then merges in five of five: scripts/sessions/r04_gate_demo.sh) and
    not much else; the comparison with power is tests/test_host_logic.py::test_c1_hard_mergeability_under_equal_search_seeds."""
    from helpers import synth_sim8_c1_hard, cp_upper, _outcome_class, merged_genomes
    st = hard_set_statistics()
    ref = json.load(open(golden_path("e2e_reference_c1_hard.json")))
    reads, labels = synth_sim8_c1_hard()
    assert ref["n_reads"] == len(reads) == 432_333 and ref["flags"] == " ".join(C1_FLAGS)
    fa = str(tmp_path / "reads.fasta")
    write_fasta(fa, reads)
    del reads
    res = []
    for seed in SEEDS:
        o = str(tmp_path / f"out{seed}")
        cmd = [sys.executable, os.path.join(ROOT, "lrbinner.py"), "reads", "-r", fa, "-o", o] + C1_FLAGS + ["--cuda", "-t", "32"]
        subprocess.run(cmd, check=True, cwd=ROOT, env=dict(os.environ, LRB_SEED=str(seed), LRB_VAE_DETERMINISTIC="1"))
        bins = [int(x) for x in open(os.path.join(o, "bins.txt")).read().split()]
        p, r, f1, nb = binning_scores(bins, labels)
        res.append({"seed": seed, "precision": p, "recall": r, "f1": f1, "bins": nb, "merged": merged_genomes(bins, labels)})
        print("C1-hard e2e (deterministic mode)", res[-1])
        shutil.rmtree(o)
    few = sum(r["bins"] < 8 for r in res)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "c1_hard_e2e_scores.json"), "w") as f:
            json.dump({"runs": res, "runs_below_8_bins": few, "statistics": {k: v for k, v in st.items() if not k.endswith("_runs")}}, f, indent=1)
    except OSError:
        pass
    ref8 = [q["f1"] for q in st["ref_runs"] if q["bins"] >= 8]
    cost = {"strain": (96.9, 97.4), "gc": (92.0, 92.6), "both": (88.9, 89.5)}
    for r in res:
        cls = _outcome_class(r)
        assert cls in ("none", "strain", "gc", "both"), r
        if cls == "none":
            assert min(abs(r["f1"] - f) for f in ref8) <= 0.5, (r, ref8)
        else:
            assert cost[cls][0] <= r["f1"] <= cost[cls][1], r
    merged = searches = 0
    for name in ("r06_c1hard_ref_recluster_s1to8.json", "r06_c1hard_ref_recluster_s1001.json"):
        for lat in json.load(open(os.path.join(ROOT, "profiles", name)))["latents"]:
            merged += sum(bool(q["merged"]) for q in lat["searches"])
            searches += len(lat["searches"])
    p_up = cp_upper(merged, searches)
    q = next(k for k in range(len(res) + 1) if _binom_sf(k, len(res), p_up) <= 0.01)
    print(f"reference latents: {merged} of {searches} searches merge a pair, upper 95 % bound {p_up:.3f} -> at most {q} of {len(res)} runs below eight bins")
    assert few <= q, (few, q, p_up, res)
