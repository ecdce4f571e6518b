#!/usr/bin/env python3
"""BASELINE config C5 at full size on one MI355X: `lrbinner.py contigs` on 500 k synthetic contigs
(k = 4, VAE, HDBSCAN over the fragments).  Per-stage wall times from the log.
python scripts/c5_full.py [n_contigs] > gpurun_out/c5_full.json"""
import json, os, re, subprocess, sys, tempfile, time
from datetime import datetime
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

n_contigs = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
rng = np.random.default_rng(5)
letters = np.frombuffer(b"ACGT", dtype=np.uint8)
n_genomes, glen = 10, 4_000_000
genomes = []
for g in range(n_genomes):
    p = rng.dirichlet(np.full(4, 6.0))   # one base composition per genome (as tests/test_gpu_c5.py)
    genomes.append(letters[(rng.random(glen)[:, None] > np.cumsum(p)[None, :]).sum(1).clip(0, 3)])
cov = np.array([5, 7, 9, 12, 16, 21, 28, 37, 48, 60], dtype=np.float64)
t0 = time.time()
with tempfile.TemporaryDirectory(dir="/dev/shm") as tmp:
    contigs, reads = os.path.join(tmp, "contigs.fasta"), os.path.join(tmp, "reads.fasta")
    origin = rng.choice(n_genomes, size=n_contigs, p=cov / cov.sum())
    lens = np.clip(rng.lognormal(8.6, 0.6, n_contigs).astype(np.int64), 1500, 60000)
    with open(contigs, "wb") as f:
        for i in range(n_contigs):
            s = int(rng.integers(0, glen - lens[i]))
            f.write(b">contig_%d\n" % i); f.write(genomes[origin[i]][s:s + lens[i]].tobytes()); f.write(b"\n")
    n_reads, L = 200_000, 8000
    rorigin = rng.choice(n_genomes, size=n_reads, p=cov / cov.sum())
    with open(reads, "wb") as f:
        for i in range(n_reads):
            s = int(rng.integers(0, glen - L))
            f.write(b">r%d\n" % i); f.write(genomes[rorigin[i]][s:s + L].tobytes()); f.write(b"\n")
    gen_s = time.time() - t0
    out = os.path.join(tmp, "out")
    cmd = [sys.executable, os.path.join(ROOT, "lrbinner.py"), "contigs", "-r", reads, "-c", contigs, "-o", out, "-k", "4",
           "--ae-dims", "8", "--ae-epochs", "200", "--cuda", "-t", "32"]
    t1 = time.time()
    if os.environ.get("C5_PROFILE"):   # the same run in this process under cProfile: where the host side spends its time
        import cProfile, io, pstats
        os.environ["LRB_SEED"] = "5"
        import lrbinner
        pr = cProfile.Profile()
        pr.enable()
        try:
            lrbinner.main(cmd[2:])
        except SystemExit:
            pass
        pr.disable()
        st = io.StringIO()
        pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(60)
        print(f"wall {time.time() - t1:.2f} s", file=sys.stderr)
        print(st.getvalue()[:12000], file=sys.stderr)
        sys.exit(0)
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, env=dict(os.environ, LRB_SEED="5"))
    wall = time.time() - t1
    if r.returncode != 0:
        print(r.stderr[-3000:]); sys.exit(1)
    stamps = []
    for line in open(os.path.join(out, "LRBinner.log")):
        m = re.match(r"(\d{4}-\d\d-\d\d \d\d:\d\d:\d\d,\d{3}) - \w+ - (.*)", line)
        if m:
            stamps.append((datetime.strptime(m.group(1), "%Y-%m-%d %H:%M:%S,%f"), m.group(2).strip()))
    gaps = [{"message": b[1][:80], "seconds_since_previous": round((b[0] - a[0]).total_seconds(), 2)}
            for a, b in zip(stamps, stamps[1:]) if (b[0] - a[0]).total_seconds() >= 0.2 and not b[1].startswith("Epoch")]
    lat = np.load(os.path.join(out, "latent.npy"))
    if os.environ.get("C5_SAVE_LATENT"):   # the first 200 k fragment latents, for the sklearn label fixture (tests/golden/make_golden_hdbscan.py)
        np.save(os.environ["C5_SAVE_LATENT"], lat[:200_000].astype(np.float32))
    rows = [l.split("\t") for l in open(os.path.join(out, "bins.txt")).read().splitlines()]
    # purity of the bins against the genome every contig was cut from
    by_bin = {}
    for cid, b in rows:
        by_bin.setdefault(b, []).append(origin[int(cid.split("_")[1])])
    pure = sum(np.bincount(v).max() for v in by_bin.values())
    # (every stage's product of this configuration is asserted in tests/test_gpu_c5.py)
    per = np.where(lens >= 5000, -(-lens // 2500) + 1, 1)
    assert lat.shape == (int(per.sum()), 8) and np.isfinite(lat).all()
    assert len(rows) > 0.5 * n_contigs and pure / max(len(rows), 1) > 0.9
    res = {"n_contigs": n_contigs, "contig_bases_GB": round(os.path.getsize(contigs) / 1e9, 2), "n_fragments": int(lat.shape[0]),
           "latent_dims": int(lat.shape[1]), "wall_s": round(wall, 1), "contigs_binned_per_s": round(n_contigs / wall),
           "log_gaps": gaps, "contigs_with_a_bin": len(rows), "bins": len(by_bin),
           "purity_of_binned_contigs": round(float(pure) / max(len(rows), 1), 4), "generate_s": round(gen_s, 1)}
    print(json.dumps(res, indent=1))
