"""The commands that START several ranks, rehearsed on the one MI355X a test box has: RCCL refuses two ranks on a
device, so the collective goes through gloo (LRB_BENCH_BACKEND / LRB_DIST_BACKEND) while everything else -- the
child job, the shards, the kernels, fold / all-reduce / expand, stitching, checkpoints -- is the real path.

    python bench.py --gpus 2                         the driver's command: n_gpus 2, the collective seen by 2 ranks
    LRB_GPUS=2 lrbinner.py reads ...                 SURVEY 8e behind the reference's CLI (lrbinner.py:12-203,
                                                     pipelines.py:242-368): stages 1_1 / 1_2 / 2_1 sharded, logged
                                                     with the reference's parameters, --resume skips them
"""
import json
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

from helpers import ROOT, gz_bytes, golden_path, synth_metagenome, write_fasta
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _bench(tmp_path, gpus, extra_args, timeout=1500):
    """`python bench.py --gpus N ...` with gloo between the ranks of ONE GPU -> (the line rank 0 printed, the detail file)."""
    detail = str(tmp_path / f"detail{gpus}.json")
    env = dict(os.environ, LRB_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", LRB_BENCH_DETAIL=detail)
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus)] + extra_args
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 prints, once
    assert len(lines[0]) < 8192                        # the contract line fits a record that keeps 8 KB of tail
    return json.loads(lines[0]), json.load(open(detail))["detail"]


def test_bench_gpus_2_starts_two_ranks(tmp_path):
    line, detail = _bench(tmp_path, 2, ["--steps", "3", "--warmup", "1", "--clock-ramp-ms", "0", "--reads", "50000",
                                        "--c4-reads", "60000", "--no-cpu-baseline", "--no-traffic"])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    assert not {"c4_phases", "roofline_stages", "extra", "vae_step"} & set(line)     # in the detail file, not in the line
    c4 = detail["c4_phases"]
    assert "error" not in c4, c4
    assert c4["world_size_seen_by_rccl"] == 2 and c4["allreduce"] == "half"
    # (K2 tallies into the canonical half: no fold before the all-reduce)
    assert {"allreduce_ms", "expand_ms", "k1_k4_ms", "k2_ms", "k3_ms"} <= set(c4["phases_ms_max_over_ranks"])
    # both routes of the coverage phase, named, timed through the product's objects (lrbinner_amd.dist.HipCompute): the
    # library's defaults head the block, the kept-list route stands beside it
    assert set(c4["routes"]) == {"default", "kept_lists"} and c4["reads_per_s_route"] == "default"
    assert c4["routes"]["default"]["groups_with_kept_lists"] == 0 and c4["routes"]["kept_lists"]["groups_with_kept_lists"] >= 1
    assert c4["reads_per_s"] == c4["routes"]["default"]["reads_per_s"] > 0 and c4["routes"]["kept_lists"]["reads_per_s"] > 0
    assert "HipCompute" in c4["timed_through"]
    assert line["value"] > 0 and "error" not in (detail.get("extra") or {})
    # the collective's time, its bus bandwidth and SURVEY 8(e)'s cost model beside it
    assert c4["allreduce_ms"] > 0 and c4["allreduce_busbw_GBps"] > 0 and c4["allreduce_bytes"] == 2 << 30
    cm = c4["allreduce_cost_model"]
    assert cm["direct_ms"] == pytest.approx(2 * (2 ** 31 / 2) / 75e9 * 1e3) and cm["ring_ms"] == cm["direct_ms"]  # P = 2
    assert "numa_pin_rank0" in line["config"]
    # ... and what a record that keeps only `roofline` still holds: every stage's fraction, the C4-shaped rank
    rf = line["roofline"]
    assert {"k1_k4", "k1_k5", "k2", "k3_default", "k3_kept_lists", "k4_seed_hist", "k5_gauss", "vae_encode", "k6_core", "k6_mst"} <= set(rf["stages"])
    for st in rf["stages"].values():
        assert set(st) == {"frac", "kernel_ms", "traffic_ratio", "bound"} and st["frac"] > 0 and st["kernel_ms"] > 0
    cr = rf["c4_rank"]
    assert cr["world_size_seen_by_rccl"] == 2 and cr["allreduce_bytes"] == 2 << 30 and cr["allreduce_ms"] > 0
    assert cr["default_reads_per_s"] == pytest.approx(c4["reads_per_s"], rel=1e-3) and cr["kept_reads_per_s"] > 0 and cr["with_text_reads_per_s"] > 0
    assert all(len(v) == 2 for v in cr["phases_ms_per_rank"].values()) and {"k2_ms", "k3_ms", "allreduce_ms"} <= set(cr["phases_ms_per_rank"])
    for k_, v in cr["phases_ms_per_rank"].items():      # the maxima are the maxima of what the ranks reported
        assert max(v) == pytest.approx(cr["phases_ms_max_over_ranks"][k_], rel=2e-3, abs=2e-3)


def test_bench_gpus_8_rehearsal_on_one_gpu(tmp_path):
    """The driver's eight-rank command before there is an eight-GPU node: `python bench.py --gpus 8`, the ranks' collectives
    through gloo, every rank's kernels on the one MI355X (the times mean nothing; the code path is the one RCCL will
    run): eight shards, eight canonical halves, ONE all-reduce of 2 GiB seen by eight ranks, per-rank phase times."""
    line, detail = _bench(tmp_path, 8, ["--steps", "2", "--warmup", "1", "--clock-ramp-ms", "0", "--reads", "20000",
                                        "--c4-reads", "20000", "--no-cpu-baseline", "--no-traffic", "--no-extra"], timeout=2400)
    assert line["n_gpus"] == 8 and line["scaling"] == "weak" and line["value"] > 0
    cr = line["roofline"]["c4_rank"]
    assert "error" not in cr, cr
    assert cr["world_size_seen_by_rccl"] == 8 and cr["allreduce"] == "half" and cr["allreduce_bytes"] == 2 << 30
    assert cr["allreduce_busbw_GBps"] == pytest.approx((2 << 30) / (cr["allreduce_ms"] * 1e-3) / 1e9 * 2 * 7 / 8, rel=2e-3)
    assert all(len(v) == 8 and min(v) > 0 for k_, v in cr["phases_ms_per_rank"].items())
    assert cr["allreduce_model_ms"]["direct_ms"] == pytest.approx(2 * (2 ** 31 / 8) / 75e9 * 1e3, rel=1e-3)
    c4 = detail["c4_phases"]
    assert c4["routes"]["default"]["reads_per_s"] > 0 and c4["routes"]["kept_lists"]["reads_per_s"] > 0


def test_bench_rank_that_dies_ends_the_job_in_minutes():
    """A rank that leaves before the first collective (LRB_BENCH_FAIL_RANK): the job ends with a non-zero status well
    inside the collective timeout instead of sitting in a barrier for torch's default half hour."""
    import time
    env = dict(os.environ, LRB_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", LRB_BENCH_FAIL_RANK="1",
               LRB_COLLECTIVE_TIMEOUT_S="60")
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--clock-ramp-ms", "0", "--reads", "20000", "--no-c4", "--no-extra", "--no-cpu-baseline", "--no-traffic"]
    t0 = time.time()
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and time.time() - t0 < 300, (r.returncode, time.time() - t0)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]      # no result line from a broken job


def test_bench_refuses_gpus_that_are_not_there():
    """--gpus 64 on this box (RCCL backend): exit 2 and a message, nothing started."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RANK", None)
    env.pop("LRB_BENCH_BACKEND", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "GPU(s)" in r.stderr, (r.returncode, r.stderr[-500:])


def _profile_files(out):
    return {f: open(os.path.join(out, "profiles", f), "rb").read() for f in ("com_profs", "cov_profs")}


def test_cli_reads_on_two_ranks_and_resume(tmp_path):
    """LRB_GPUS=2 lrbinner.py reads: the three profile files are the oracle's bytes (and the table the oracle's
    sparse table), the checkpoints carry the reference's stage parameters, the run goes on through npy, VAE and
    clustering on rank 0, and --resume starts no rank again."""
    reads, labels = synth_metagenome()
    fa = str(tmp_path / "reads.fasta")
    write_fasta(fa, reads)
    out = str(tmp_path / "out")
    env = dict(os.environ, LRB_GPUS="2", LRB_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", LRB_SEED="3")
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "lrbinner.py"), "reads", "-r", fa, "-o", out, "-k", "4", "-bc", "10",
           "-bs", "8", "--ae-dims", "4", "--ae-epochs", "20", "-bit", "0", "-mbs", "200", "--cuda", "-t", "8"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    log = open(os.path.join(out, "LRBinner.log")).read()
    assert "Profile stages on 2 GPUs" in log
    # bytes: the oracle's
    buf, offs = orc.concat(reads)
    counts, totals = orc.count_kmers(buf, offs, 4)
    got = _profile_files(out)
    assert got["com_profs"] == orc.format_com(orc.com_profile(counts, totals))
    keys, cnts = orc.k15_sparse(buf, offs)
    hist, sums = orc.cov_hist(buf, offs, keys, cnts, 8, 10)
    assert got["cov_profs"] == orc.format_cov(orc.cov_profile(hist, sums))
    tpath = os.path.join(out, "profiles", "15mers-counts")
    assert os.path.getsize(tpath) == 8 + 4 * 4 ** 15
    table = np.memmap(tpath, dtype=np.uint32, mode="r", offset=8)
    assert np.array_equal(table[keys], cnts) and int(np.count_nonzero(table)) == len(keys)
    del table
    # the value side-cars were stitched too, and stage 3_1 read them: npy == float(token) of the text
    com = np.load(os.path.join(out, "profiles/com_profs.npy"))
    assert com.dtype == np.float64 and com.shape == (len(reads), 136)
    first = np.array([float(t) for t in got["com_profs"].split(b"\n", 1)[0].split()])
    assert np.array_equal(com[0], first)
    cp = pickle.load(open(os.path.join(out, "checkpoints"), "rb"))
    assert cp["1_1"] == [fa, 4] and cp["1_2"] == [fa] and cp["2_1"] == [fa, 8, 10] and cp["3_1"] == ["numpy"]
    for f in ("model.pt", "latent.npy", "bins.txt", "lengths.txt", "binning_result.pkl"):
        assert os.path.exists(os.path.join(out, f)), f
    assert len(open(os.path.join(out, "bins.txt")).read().split()) == len(reads)
    # --resume: the profile stages are skipped, no rank is started, the files stay as they are
    before = {f: os.stat(os.path.join(out, "profiles", f)).st_mtime_ns for f in ("com_profs", "cov_profs", "15mers-counts")}
    r = subprocess.run(cmd + ["--resume"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    log2 = open(os.path.join(out, "LRBinner.log")).read()[len(log):]
    assert "Profile stages on" not in log2
    for msg in ("K-mer vectors already computed", "15-mers already counted", "Already computed 15-mer profiles complete",
                "Numpy arrays already computed", "VAE already trained"):
        assert msg in log2, msg
    after = {f: os.stat(os.path.join(out, "profiles", f)).st_mtime_ns for f in before}
    assert before == after
    os.remove(tpath)


def _oracle_profiles(reads, k, bs, bc):
    buf, offs = orc.concat(reads)
    counts, totals = orc.count_kmers(buf, offs, k)
    keys, cnts = orc.k15_sparse(buf, offs)
    hist, sums = orc.cov_hist(buf, offs, keys, cnts, bs, bc)
    return orc.format_com(orc.com_profile(counts, totals)), orc.format_cov(orc.cov_profile(hist, sums)), keys, cnts


def test_cli_reads_on_eight_ranks(tmp_path):
    """SURVEY 8(e) at the width of the node it is written for, rehearsed on one GPU (gloo between the ranks):
    `LRB_GPUS=8 lrbinner.py reads` -- eight shards of byte ranges in file order (count-kmers.cpp:97-123: rows in input
    order), eight canonical halves, one all-reduce, every rank writing its rows and its eighth of the table file.
      * all eight ranks busy (4 MB ranges of a 35 MB file): the oracle's profile files byte for byte, the table the
        oracle's sparse table, eight stage-stamp files whose rows add up;
      * seven ranks with NOTHING to parse (the default 64 MB range holds the whole file): the same bytes;
      * FASTQ (cannot be cut: every rank streams, batch b goes to rank b mod 8): the same bytes;
      * --resume: no rank is started again;
      * a rank that dies before the first collective: the job ends non-zero inside the collective timeout and leaves
        no file that looks like a profile (.partial names until every rank's rows are in)."""
    import time
    reads, labels = synth_metagenome()
    fa = str(tmp_path / "reads.fasta")
    write_fasta(fa, reads)
    want_com, want_cov, keys, cnts = _oracle_profiles(reads, 4, 8, 10)
    base_env = dict(os.environ, LRB_GPUS="8", LRB_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", LRB_SEED="3")
    base_env.pop("RANK", None)

    def cmd_for(reads_path, out):
        return [sys.executable, os.path.join(ROOT, "lrbinner.py"), "reads", "-r", reads_path, "-o", out, "-k", "4", "-bc", "10",
                "-bs", "8", "--ae-dims", "4", "--ae-epochs", "5", "-bit", "0", "-mbs", "200", "--cuda", "-t", "8"]

    def check(out):
        got = _profile_files(out)
        assert got["com_profs"] == want_com and got["cov_profs"] == want_cov
        tpath = os.path.join(out, "profiles", "15mers-counts")
        assert os.path.getsize(tpath) == 8 + 4 * 4 ** 15
        table = np.memmap(tpath, dtype=np.uint32, mode="r", offset=8)
        assert np.array_equal(table[keys], cnts) and int(np.count_nonzero(table)) == len(keys)
        del table
        assert not [f for f in os.listdir(os.path.join(out, "profiles")) if f.endswith(".partial") or ".spill" in f]
        meta = json.load(open(os.path.join(out, "profiles", "com_profs.q6.json")))
        assert meta["rows"] == len(reads) and meta["cols"] == 136
        assert len(open(os.path.join(out, "bins.txt")).read().split()) == len(reads)
        os.remove(tpath)

    # all eight ranks busy
    out = str(tmp_path / "out8")
    stats = str(tmp_path / "stats8")
    r = subprocess.run(cmd_for(fa, out), cwd=ROOT, env=dict(base_env, LRB_PARSE_CHUNK_BYTES=str(4 << 20), LRB_DIST_STATS=stats),
                       capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "Profile stages on 8 GPUs" in open(os.path.join(out, "LRBinner.log")).read()
    ranks = [json.load(open(f"{stats}.rank{i}.json")) for i in range(8)]
    assert [q["rank"] for q in ranks] == list(range(8)) and all(q["world"] == 8 and q["rows"] == len(reads) for q in ranks)
    assert all(q["direct_bytes"] + q["buffered_bytes"] + q["spilled_bytes"] > 0 for q in ranks)    # every rank wrote rows of its own
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump({"what": "LRB_GPUS=8 lrbinner.py reads, gloo ranks on one MI355X, 11,703 reads of 3 kb in 4 MB ranges (tests/test_gpu_multi.py)",
                   "ranks": ranks}, open(os.path.join(ROOT, "gpurun_out", "r06_dist_cli8_stamps.json"), "w"), indent=1)
    except OSError:
        pass
    # --resume: the profile stages are skipped, no rank is started
    log = open(os.path.join(out, "LRBinner.log")).read()
    r = subprocess.run(cmd_for(fa, out) + ["--resume"], cwd=ROOT, env=base_env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "Profile stages on" not in open(os.path.join(out, "LRBinner.log")).read()[len(log):]
    check(out)
    # seven ranks without a byte range
    out = str(tmp_path / "out8empty")
    r = subprocess.run(cmd_for(fa, out), cwd=ROOT, env=base_env, capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stderr[-3000:]
    check(out)
    # FASTQ: streamed by every rank, batches dealt out in turn
    fq = str(tmp_path / "reads.fastq")
    with open(fq, "wb") as f:
        for i, rd in enumerate(reads):
            f.write(b"@read%d\n" % i + rd + b"\n+\n" + b"I" * len(rd) + b"\n")
    out = str(tmp_path / "out8fq")
    r = subprocess.run(cmd_for(fq, out), cwd=ROOT, env=base_env, capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stderr[-3000:]
    check(out)
    # a dying rank
    out = str(tmp_path / "out8dead")
    t0 = time.time()
    r = subprocess.run(cmd_for(fa, out), cwd=ROOT, env=dict(base_env, LRB_DIST_FAIL_RANK="5", LRB_COLLECTIVE_TIMEOUT_S="60",
                                                            LRB_PARSE_CHUNK_BYTES=str(4 << 20)),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode != 0 and time.time() - t0 < 420, (r.returncode, time.time() - t0)
    prof = os.path.join(out, "profiles")
    left = os.listdir(prof) if os.path.isdir(prof) else []
    assert not {"com_profs", "cov_profs", "15mers-counts", "com_profs.q6.json", "cov_profs.q6.json"} & set(left), left


def test_cli_reads_under_a_launcher_two_ranks(tmp_path):
    """The same route when the USER starts the ranks (python -m torch.distributed.run ... lrbinner.py reads): rank 1
    takes its share of the profile stages and leaves, rank 0 carries on; the reference's files on edge.fasta."""
    out = str(tmp_path / "out")
    env = dict(os.environ, LRB_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    env.pop("LRB_GPUS", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29531", os.path.join(ROOT, "lrbinner.py"), "reads",
           "-r", golden_path("edge.fasta"), "-o", out, "-k", "3", "-bs", "10", "-bc", "32", "--ae-epochs", "1",
           "-bit", "0", "-mbs", "1", "-t", "2"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    # (what the clustering makes of a few dozen edge-case reads is not the point: the profile stages are)
    log = open(os.path.join(out, "LRBinner.log")).read()
    assert "Profile stages on 2 GPUs" in log and "Computing 15-mer profiles complete" in log, r.stderr[-3000:]
    assert open(f"{out}/profiles/com_profs", "rb").read() == gz_bytes("com_profs_k3.txt.gz")
    assert open(f"{out}/profiles/cov_profs", "rb").read() == gz_bytes("cov_profs_bs10_bc32.txt.gz")
    assert os.path.getsize(f"{out}/profiles/15mers-counts") == 8 + 4 * 4 ** 15
    cp = pickle.load(open(os.path.join(out, "checkpoints"), "rb"))
    assert cp["1_1"] == [golden_path("edge.fasta"), 3] and cp["2_1"] == [golden_path("edge.fasta"), 10, 32]
    os.remove(f"{out}/profiles/15mers-counts")
