#!/usr/bin/env python3
"""The tail of the multi-GPU file path, measured where only one GPU is at hand: two ranks (gloo) on ONE MI355X, n reads
of 10 kb (k = 4, bs 10, bc 32), every rank's stage stamps from lrbinner_amd.dist.profile_file_sharded
(LRB_DIST_STATS): parse + pack + K1, the layout exchange, K2, all-reduce + expand, K3, and -- the point -- what is left
after the last kernel: this rank's rows still being written, the wait in the last barrier, rank 0's side-car
descriptions, rank 0's wait for its table file.  The kernels of two ranks share one GPU here (their times mean
nothing); the host-side tail is what an 8-GPU node would see per rank.  Also checks the multi-rank files against a
one-rank run, byte for byte.  python3 scripts/dist_tail_probe.py [n=2000000] [worlds=1,2]   (round 6: worlds 1,8 ->
gpurun_out/r06_dist_tail8.json)"""
import filecmp, json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
WORLDS = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2").split(",")]
L = 10_000
rng = np.random.default_rng(1)
with tempfile.TemporaryDirectory(dir="/dev/shm") as tmp:
    fa = os.path.join(tmp, "reads.fasta")
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    block = 20000
    rows = np.empty((block, L + 1), dtype=np.uint8)
    rows[:, :L] = letters[rng.integers(0, 4, size=(block, L), dtype=np.uint8)]; rows[:, L] = 10
    with open(fa, "wb") as f:
        for s in range(0, n, block):
            rows[:, :L] = np.roll(rows[:, :L], 37, axis=1)
            for i in range(min(block, n - s)):
                f.write(b">r%d\n" % (s + i)); f.write(rows[i].tobytes())
    res = {"n_reads": n, "read_len": L, "k": 4, "bins": 32, "fasta_GB": round(os.path.getsize(fa) / 1e9, 2), "runs": {}}
    for world in WORLDS:
        out = os.path.join(tmp, f"out{world}")
        env = dict(os.environ, LRB_DIST_BACKEND="gloo", LRB_DIST_STATS=os.path.join(tmp, f"stats{world}"), MASTER_ADDR="127.0.0.1",
                   PYTHONPATH=ROOT, OMP_NUM_THREADS=str(max(2, 16 // world)))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--standalone",
               "--local-addr", "127.0.0.1", "-m", "lrbinner_amd.dist", "--reads", fa, "--output", out, "-k", "4", "-bs", "10", "-bc", "32", "-t", str(max(4, 32 // world))]
        t0 = time.time()
        subprocess.run(cmd, check=True, env=env, cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        wall = time.time() - t0
        ranks = [json.load(open(os.path.join(tmp, f"stats{world}.rank{r}.json"))) for r in range(world)]
        tail_keys = ("rows_written_after_last_kernel_s", "barrier_wait_s", "rank0_sidecar_s", "table_file_wait_s")
        res["runs"][f"world{world}"] = {
            "job_wall_s": round(wall, 2), "ranks": ranks,
            "rank0_only_after_last_kernel_s": round(ranks[0]["rank0_sidecar_s"], 4),
            "tail_s_max_over_ranks": round(max(sum(r[k] for k in tail_keys) for r in ranks), 4),
            "stage_total_s_max_over_ranks": round(max(r["total_s"] for r in ranks), 4),
        }
        print(world, json.dumps(res["runs"][f"world{world}"])[:1500], flush=True)
    top = max(WORLDS)
    same = {f: filecmp.cmp(os.path.join(tmp, "out1", "profiles", f), os.path.join(tmp, f"out{top}", "profiles", f), shallow=False)
            for f in ("com_profs", "cov_profs", "com_profs.q6", "cov_profs.q6", "15mers-counts")} if 1 in WORLDS and top > 1 else {}
    res[f"{top}_ranks_equal_one_rank"] = same
    wt = res["runs"][f"world{top}"]
    res["rank0_only_share_of_stage"] = round(wt["rank0_only_after_last_kernel_s"] / wt["stage_total_s_max_over_ranks"], 4)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", f"r06_dist_tail{top}.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "runs"}))
    assert all(same.values()), same
