#!/usr/bin/env python3
"""How often does this build merge the strain pair of the hard C1 set (tests/helpers.synth_sim8_c1_hard)?  N whole
`lrbinner.py reads` runs with the README's flags, seeds 1..N, library defaults (the VAE's batch sums by float atomics:
a seed does not fix the outcome) -> bins and F1 per seed, the count of runs below eight bins; then the test's five
seeds under LRB_VAE_DETERMINISTIC=1, each TWICE: the two runs of a seed must give byte-identical latent.npy and
bins.txt.  Written to gpurun_out/r05_c1_hard_rates.json (copied to profiles/).
python3 scripts/c1_hard_rates.py [N=50]
C1_HARD_TORCH=M: instead, M runs (seeds 1..M) with the VAE trained by THIS build's torch-module path on the GPU
(LRB_VAE_NATIVE=0: autograd, torch's Adam and BatchNorm -- the reference's arithmetic, ae_utils.py:199-241) -> are the
merges the fused step's or the method's?  Written to gpurun_out/r05_c1_hard_rates_torch.json.
C1_HARD_DET=D: instead, D seeds of the fused step under LRB_VAE_DETERMINISTIC=1 (ordered batch sums), once each ->
gpurun_out/r05_c1_hard_rates_det.json."""
import hashlib, json, os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import binning_scores, synth_sim8_c1_hard, write_fasta

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
flags = "-k 3 -bc 10 -bs 32 --ae-dims 4 --ae-epochs 200 -bit 0 -mbs 5000".split()


def sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]


with tempfile.TemporaryDirectory(dir="/dev/shm") as tmp:
    reads, labels = synth_sim8_c1_hard()
    fa = os.path.join(tmp, "reads.fasta")
    write_fasta(fa, reads)
    del reads

    def run(seed, det, torch_path=False):
        o = os.path.join(tmp, "out")
        shutil.rmtree(o, ignore_errors=True)
        env = dict(os.environ, LRB_SEED=str(seed))
        if torch_path:
            env["LRB_VAE_NATIVE"] = "0"
        if det:
            env["LRB_VAE_DETERMINISTIC"] = "1"
        else:
            env.pop("LRB_VAE_DETERMINISTIC", None)
        t0 = time.time()
        subprocess.run([sys.executable, os.path.join(ROOT, "lrbinner.py"), "reads", "-r", fa, "-o", o] + flags + ["--cuda", "-t", "32"],
                       check=True, cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        wall = time.time() - t0
        bins = [int(x) for x in open(os.path.join(o, "bins.txt")).read().split()]
        p, r, f1, nb = binning_scores(bins, labels)
        return {"seed": seed, "bins": nb, "f1": f1, "precision": p, "recall": r, "wall_s": round(wall, 1),
                "latent_sha": sha(os.path.join(o, "latent.npy")), "bins_sha": sha(os.path.join(o, "bins.txt"))}

    M = int(os.environ.get("C1_HARD_TORCH", "0"))
    if M:
        out = {"dataset": "helpers.synth_sim8_c1_hard()", "n_reads": len(labels), "flags": " ".join(flags),
               "vae": "torch modules on the GPU (LRB_VAE_NATIVE=0)", "runs": []}
        path = os.path.join(ROOT, "gpurun_out", "r05_c1_hard_rates_torch.json")
        for seed in range(1, M + 1):
            out["runs"].append(run(seed, False, torch_path=True))
            print("torch path", out["runs"][-1], flush=True)
            out["runs_below_8_bins"] = sum(r["bins"] < 8 for r in out["runs"])
            json.dump(out, open(path, "w"), indent=1)
        sys.exit(0)
    D = int(os.environ.get("C1_HARD_DET", "0"))
    if D:   # D seeds under LRB_VAE_DETERMINISTIC=1, once each: does the ORDER of the batch sums move the merge rate?
        out = {"dataset": "helpers.synth_sim8_c1_hard()", "n_reads": len(labels), "flags": " ".join(flags),
               "vae": "fused step, LRB_VAE_DETERMINISTIC=1", "runs": []}
        path = os.path.join(ROOT, "gpurun_out", "r05_c1_hard_rates_det.json")
        for seed in range(1, D + 1):
            out["runs"].append(run(seed, True))
            print("deterministic", {k: out["runs"][-1][k] for k in ("seed", "bins", "f1")}, flush=True)
            out["runs_below_8_bins"] = sum(r["bins"] < 8 for r in out["runs"])
            json.dump(out, open(path, "w"), indent=1)
        sys.exit(0)
    out = {"dataset": "helpers.synth_sim8_c1_hard()", "n_reads": len(labels), "flags": " ".join(flags), "default_mode": [], "deterministic_mode": []}
    path = os.path.join(ROOT, "gpurun_out", "r05_c1_hard_rates.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    for seed in range(1, N + 1):
        out["default_mode"].append(run(seed, False))
        print("default", out["default_mode"][-1], flush=True)
        few = sum(r["bins"] < 8 for r in out["default_mode"])
        out["default_runs"] = len(out["default_mode"]); out["default_runs_below_8_bins"] = few
        json.dump(out, open(path, "w"), indent=1)
    for seed in (1, 2, 3, 4, 5):
        a, b = run(seed, True), run(seed, True)
        a["repeat_identical"] = a["latent_sha"] == b["latent_sha"] and a["bins_sha"] == b["bins_sha"]
        out["deterministic_mode"].append(a)
        print("deterministic", a, flush=True)
    out["deterministic_all_repeats_identical"] = all(r["repeat_identical"] for r in out["deterministic_mode"])
    out["deterministic_runs_below_8_bins"] = sum(r["bins"] < 8 for r in out["deterministic_mode"])
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if not isinstance(v, list)}))
