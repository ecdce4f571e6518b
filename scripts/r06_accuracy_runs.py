#!/usr/bin/env python3
"""Round 6: whole `lrbinner.py reads` runs of this build on a C1 stand-in, with what two-of-twenty-five events cannot show.

Per run (README flags, seeds 1..N, library defaults): bins, F1, WHICH genomes share a bin, the continuous separation
statistics of latent.npy (tests/helpers.latent_pair_stats: d' and valley ratio per genome pair), and the SAME latent.npy
clustered again under R other search seeds (this build's cluster search, which finds the reference's clusters seed by
seed -- tests/test_gpu_sim8.py::test_sim8_reference_latents_through_this_clustering): is an outcome a property of the
trained latents or of the search's random start?

    python3 scripts/r06_accuracy_runs.py c1hard|c1 [N=60] [R=3]      -> gpurun_out/r06_<set>_runs.json
    R06_VAE=torch    the torch-module VAE on the GPU (LRB_VAE_NATIVE=0)  -> ..._runs_torch.json
    R06_LATENTS=DIR  no runs: every *.npy in DIR (reference-trained latents shipped from the build container) through
                     the search under seeds 1..R                       -> gpurun_out/r06_<set>_ref_recluster.json"""
import json, os, random, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import C1H_PAIRS, C1_PAIRS, binning_scores, latent_pair_stats, synth_sim8_c1, synth_sim8_c1_hard, write_fasta

SET = sys.argv[1] if len(sys.argv) > 1 else "c1hard"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
R = int(sys.argv[3]) if len(sys.argv) > 3 else 3
# R06_SEARCH_SEEDS=a,b,c: the search seeds of every re-clustering (default: 1..R for shipped latents, 1001..1000+R for runs)
SEARCH_SEEDS = [int(x) for x in os.environ["R06_SEARCH_SEEDS"].split(",")] if os.environ.get("R06_SEARCH_SEEDS") else None
FIRST_K = int(os.environ.get("R06_FIRST_STEP", "0"))     # candidates of the first-step statistic (0: not computed)
MBS = 5000
flags = f"-k 3 -bc 10 -bs 32 --ae-dims 4 --ae-epochs 200 -bit 0 -mbs {MBS}".split()
PAIRS = C1H_PAIRS if SET == "c1hard" else C1_PAIRS
VAE = os.environ.get("R06_VAE", "native")


def outcome(bins, labels):
    """bins, F1 and the genome groups that ended in one bin (majority genome of a bin = its owner; a genome whose
    reads mostly sit in a bin owned by another genome is merged into it)."""
    p, r, f1, nb = binning_scores(bins, labels)
    bins, labels = np.asarray(bins), np.asarray(labels)
    home = {}
    for g in sorted(set(labels.tolist())):
        vals, cnt = np.unique(bins[labels == g], return_counts=True)
        home[g] = int(vals[np.argmax(cnt)])
    groups = {}
    for g, b in home.items():
        groups.setdefault(b, []).append(g)
    merged = sorted(tuple(v) for v in groups.values() if len(v) > 1)
    return {"bins": nb, "f1": round(f1, 4), "precision": round(p, 4), "recall": round(r, 4), "merged": [list(m) for m in merged]}


def recluster(latent, labels, seeds):
    """this build's cluster search alone on one latent matrix under random.seed(s): clusters above -mbs and who merged
    (reads no cluster took are left out, as the search leaves them)."""
    from lrbinner_amd import cluster_utils
    be = cluster_utils.HipBackend()
    res = []
    for s in seeds:
        random.seed(s)
        clusters = cluster_utils.cluster_points(latent, 0, MBS, backend=be)
        big = [np.fromiter(v, dtype=np.int64, count=len(v)) for v in clusters.values() if len(v) > MBS]
        own = {}
        for i, idx in enumerate(big):
            cnt = np.bincount(labels[idx], minlength=8)
            for g in np.flatnonzero(cnt > 0.5 * np.bincount(labels, minlength=8)):
                own.setdefault(i, []).append(int(g))
        res.append({"search_seed": s, "clusters": len(big), "merged": sorted(v for v in own.values() if len(v) > 1)})
    return res


def first_step(latent, labels, K=400, seed=12345):
    """What the exhaustive search's FIRST accepted candidate does, as a rate over K candidates drawn at random from all
    reads (cluster_utils.get_cluster_center on the full matrix, as cluster_points calls it for its first candidates):
    the share of candidates that yield a cluster at all, and among those the share whose cluster (distance <= tail) holds
    more than half of TWO genomes -- per pair.  A continuous, per-latent estimate of how mergeable the latents are: the
    whole search's 7-bin outcomes are draws from (a later-step version of) this."""
    from lrbinner_amd import cluster_utils
    be = cluster_utils.HipBackend()
    be.load(latent)
    rng = np.random.default_rng(seed)
    cand = rng.choice(len(labels), K, replace=False)
    sizes = np.bincount(labels, minlength=8)
    random.seed(seed)
    out = {"candidates": int(K), "accepted": 0, "merged": {}}
    for c in cand.tolist():
        _, dist, _, _, tail = cluster_utils.get_cluster_center(be, int(c))
        if not tail:
            continue
        out["accepted"] += 1
        inside = np.bincount(labels[dist <= tail], minlength=8)
        held = [int(g) for g in np.flatnonzero(inside > 0.5 * sizes)]
        if len(held) > 1:
            key = "+".join(map(str, held))
            out["merged"][key] = out["merged"].get(key, 0) + 1
    out["merged_rate"] = round(sum(out["merged"].values()) / max(out["accepted"], 1), 5)
    out["strain_rate"] = round(sum(v for k_, v in out["merged"].items() if "6" in k_.split("+") and "7" in k_.split("+")) / max(out["accepted"], 1), 5)
    return out


def main():
    reads, labels = (synth_sim8_c1_hard if SET == "c1hard" else synth_sim8_c1)()
    labels = np.asarray(labels)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    lat_dir = os.environ.get("R06_LATENTS")
    if lat_dir:
        del reads
        path = os.path.join(ROOT, "gpurun_out", f"r06_{SET}_ref_recluster{os.environ.get('R06_TAG', '')}.json")
        out = {"dataset": SET, "search": f"this build's cluster_points(latent, 0, {MBS}) under random.seed(1..{R})", "latents": []}
        for name in sorted(os.listdir(lat_dir)):
            if not name.endswith(".npy"):
                continue
            lat = np.load(os.path.join(lat_dir, name)).astype(np.float32)
            seeds = SEARCH_SEEDS or range(1, R + 1)
            if os.environ.get("R06_OWN_SEED"):   # ref_latent_s{n}.npy under search seed n: the reference's own whole-run outcome must come out
                seeds = [int(name.split("_s")[-1].split(".")[0])]
            rec = {"file": name, "searches": recluster(lat, labels, seeds)}
            if FIRST_K:
                rec["first_step"] = first_step(lat, labels, FIRST_K)
            out["latents"].append(rec)
            print(name, [(q["clusters"], q["merged"]) for q in rec["searches"]], rec.get("first_step"), flush=True)
            json.dump(out, open(path, "w"), indent=1)
        return
    with tempfile.TemporaryDirectory(dir="/dev/shm") as tmp:
        fa = os.path.join(tmp, "reads.fasta")
        write_fasta(fa, reads)
        del reads
        out = {"dataset": f"helpers.synth_sim8_{'c1_hard' if SET == 'c1hard' else 'c1'}()", "n_reads": int(len(labels)), "flags": " ".join(flags),
               "vae": "fused HIP step (library default)" if VAE == "native" else "torch modules on the GPU (LRB_VAE_NATIVE=0)",
               "recluster_seeds": R, "runs": []}
        path = os.path.join(ROOT, "gpurun_out", f"r06_{SET}_runs{'' if VAE == 'native' else '_torch'}{os.environ.get('R06_TAG', '')}.json")
        for seed in range(1, N + 1):
            o = os.path.join(tmp, "out")
            shutil.rmtree(o, ignore_errors=True)
            env = dict(os.environ, LRB_SEED=str(seed))
            env.pop("LRB_VAE_DETERMINISTIC", None)
            if VAE != "native":
                env["LRB_VAE_NATIVE"] = "0"
            t0 = time.time()
            subprocess.run([sys.executable, os.path.join(ROOT, "lrbinner.py"), "reads", "-r", fa, "-o", o] + flags + ["--cuda", "-t", "32"],
                           check=True, cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            wall = time.time() - t0
            bins = [int(x) for x in open(os.path.join(o, "bins.txt")).read().split()]
            rec = {"seed": seed, "wall_s": round(wall, 1)}
            rec.update(outcome(bins, labels))
            lat = np.load(os.path.join(o, "latent.npy"))
            rec["pairs"] = {k: {q: round(x, 5) for q, x in v.items()} for k, v in latent_pair_stats(lat, labels, PAIRS).items()}
            if R:
                rec["searches"] = recluster(lat, labels, SEARCH_SEEDS or range(1001, 1001 + R))
            if FIRST_K:
                rec["first_step"] = first_step(lat, labels, FIRST_K)
            out["runs"].append(rec)
            print({k: v for k, v in rec.items() if k != "pairs"}, "strain d'" if SET == "c1hard" else "",
                  rec["pairs"].get("strain", {}).get("dprime"), flush=True)
            out["runs_below_8_bins"] = sum(r["bins"] < 8 for r in out["runs"])
            json.dump(out, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
