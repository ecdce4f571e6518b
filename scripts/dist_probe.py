#!/usr/bin/env python3
"""Wall time of the sharded file driver (lrbinner_amd.dist) on one rank, against the same
stages through the single-GPU runner shims.  python scripts/dist_probe.py [n_reads]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
L = 10_000
rng = np.random.default_rng(1)
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
    fa = os.path.join(tmp, "reads.fasta")
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    with open(fa, "wb") as f:
        for s in range(0, n, 20000):
            m = min(20000, n - s)
            seqs = letters[rng.integers(0, 4, size=(m, L), dtype=np.uint8)]
            rows = np.empty((m, L + 1), dtype=np.uint8); rows[:, :L] = seqs; rows[:, L] = 10
            for i in range(m):
                f.write(b">r%d\n" % (s + i)); f.write(rows[i].tobytes())
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in (3, 4):
        out = os.path.join(tmp, f"out_k{k}")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
               "--master-addr", "127.0.0.1", "--master-port", "29541", "-m", "lrbinner_amd.dist",
               "--reads", fa, "--output", out, "-k", str(k), "-bs", "10", "-bc", "32", "-t", "16", "--no-table-file"]
        t0 = time.time(); subprocess.run(cmd, check=True, cwd=ROOT, env=env); dt = time.time() - t0
        print(f"dist driver k={k}: {dt:.2f} s wall incl. process start, torch import and RCCL init "
              f"({n} reads, com {os.path.getsize(out + '/profiles/com_profs')>>20} MB, cov {os.path.getsize(out + '/profiles/cov_profs')>>20} MB)", flush=True)
    # in-process timing of the same function (no process start-up)
    from lrbinner_amd import dist as ld
    comp = ld.HipCompute(0)
    for k in (3, 4):
        out = os.path.join(tmp, f"inproc_k{k}")
        t0 = time.time(); ld.profile_file_sharded(fa, out, k, 10, 32, 16, comp, write_table=False); dt = time.time() - t0
        print(f"profile_file_sharded k={k} in process: {dt:.3f} s = {n/dt:,.0f} reads/s", flush=True)
    a = open(os.path.join(tmp, "out_k3/profiles/cov_profs"), "rb").read()
    b = open(os.path.join(tmp, "inproc_k3/profiles/cov_profs"), "rb").read()
    assert a == b
