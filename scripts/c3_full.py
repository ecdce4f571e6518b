#!/usr/bin/env python3
"""BASELINE config C3 at full size on one MI355X: 5 M synthetic 10 kb reads (50 GB of FASTA in
tmpfs), k = 4 composition + 15-mer table + coverage histograms through the runner shims
(file in -> profile files out), text -> npy, VAE encode of the 5 M x 168 profile matrix.
python scripts/c3_full.py [n_reads] > gpurun_out/c3_full.json"""
import json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
L = 10_000
rng = np.random.default_rng(1)
res = {"n_reads": n, "read_len": L, "k": 4, "bin_size": 10, "bins": 32}
with tempfile.TemporaryDirectory(dir="/dev/shm") as tmp:
    fa = os.path.join(tmp, "reads.fasta")
    t0 = time.time()
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    block = 20000
    seqs = letters[rng.integers(0, 4, size=(block, L), dtype=np.uint8)]
    rows = np.empty((block, L + 1), dtype=np.uint8); rows[:, :L] = seqs; rows[:, L] = 10
    with open(fa, "wb") as f:
        for s in range(0, n, block):
            m = min(block, n - s)
            # fresh bases for a tenth of every block, the rest rotated: distinct reads, cheap to make
            rows[: block // 10, :L] = letters[rng.integers(0, 4, size=(block // 10, L), dtype=np.uint8)]
            rows[:, :L] = np.roll(rows[:, :L], 37, axis=1)
            rows[:] = np.roll(rows, 1, axis=0)
            for i in range(m):
                f.write(b">r%d\n" % (s + i)); f.write(rows[i].tobytes())
    res["fasta_GB"] = round(os.path.getsize(fa) / 1e9, 2)
    res["generate_s"] = round(time.time() - t0, 1)
    from lrbinner_amd import runners_utils as ru, pipelines, ae_utils, device as lrb
    out = os.path.join(tmp, "out")
    warm = os.path.join(tmp, "warm.fasta")
    with open(warm, "wb") as f:
        f.write(b">w\n" + b"ACGT" * 100 + b"\n")
    ru.run_kmers(warm, os.path.join(tmp, "warm_out"), 4, 2)
    ru.release_resident()
    total = 0.0
    class _AllDue:   # (a fresh run: every stage is due)
        def should_run_step(self, stage, params):
            return True

    def kmers():   # as the pipeline runs the stage: the conversion of com_profs starts behind it
        ru.run_kmers(fa, out, 4, 32)
        pipelines._convert_early(_AllDue(), "3_1", out, "com_profs")

    def counts():
        ru.run_15mer_counts(fa, out, 32, coverage_bins=32)   # as the pipeline calls it

    for name, fn in (("run_kmers_k4", kmers),
                     ("run_15mer_counts", counts),
                     ("run_15mer_vecs", lambda: ru.run_15mer_vecs(fa, out, 10, 32, 32)),
                     ("text_to_npy", lambda: pipelines._profiles_to_npy(out))):
        t0 = time.time(); fn(); dt = time.time() - t0
        total += dt
        res[name] = {"s": round(dt, 2), "reads_per_s": round(n / dt)}
    res["profiles_total_s"] = round(total, 2)
    res["profiles_reads_per_s"] = round(n / total)
    # (stage 3_1 hands the arrays on in memory; the two .npy files reach the disk on writer threads, under the VAE stage)
    t0 = time.time()
    from lrbinner_amd import _npcache as _npc
    _npc.finish()
    res["npy_files_complete_after_another_s"] = round(time.time() - t0, 2)
    for f in ("com_profs", "cov_profs", "15mers-counts", "com_profs.npy", "cov_profs.npy"):
        res[f"size_GB:{f}"] = round(os.path.getsize(os.path.join(out, "profiles", f)) / 1e9, 2)
    # what the files must be whatever their content: fixed-width %f rows, the table's header + 2^30 counters
    # (the device-resident form of this configuration is asserted value by value in
    # tests/test_gpu_parity.py::test_c3_full_size_device_resident)
    assert os.path.getsize(os.path.join(out, "profiles", "com_profs")) == n * (9 * 136 + 1)
    assert os.path.getsize(os.path.join(out, "profiles", "cov_profs")) == n * 9 * 32
    assert os.path.getsize(os.path.join(out, "profiles", "15mers-counts")) == 8 + 4 * 4 ** 15
    # VAE encode (random-initialised network of the C3 shape, as bench rules allow: no checkpoint offline)
    import torch
    from lrbinner_amd.vae_native import NativeTrainer
    t0 = time.time()
    from lrbinner_amd import _npcache  # as vae_encode does: the arrays stage 3_1 has just written
    comp = _npcache.load(os.path.join(out, "profiles/com_profs.npy")); cov = _npcache.load(os.path.join(out, "profiles/cov_profs.npy"))
    assert comp.shape == (n, 136) and cov.shape == (n, 32)
    assert np.allclose(comp[:: max(1, n // 1000)].sum(1), 1.0, atol=136e-6) and np.allclose(cov[:: max(1, n // 1000)].sum(1), 1.0, atol=1e-2)
    data = ae_utils.make_data(cov, comp, "cuda")
    res["load_scale_upload_s"] = round(time.time() - t0, 2)
    vae = ae_utils.VAE(cov.shape[1], comp.shape[1], latent_dims=8, hidden_layers=[128, 128], device="cuda")
    ctx = lrb.Context(0, use_torch_stream=True)
    w = ae_utils.h_params[str(comp.shape[1])]
    tr = NativeTrainer(ctx, vae, 8192, [w["e_cov_weight"], w["e_comp_weight"], w["kld_weight"]])
    tr.push()
    tr.encode(data[:8192]); torch.cuda.synchronize()
    t0 = time.time(); mu = tr.encode(data); torch.cuda.synchronize(); dt = time.time() - t0
    res["vae_encode_fused"] = {"s": round(dt, 3), "rows_per_s": round(n / dt)}
    os.environ["LRB_VAE_NATIVE"] = "0"
    vae.encode(data[:8192])
    t0 = time.time(); ref = vae.encode(data); dt = time.time() - t0
    res["vae_encode_torch_incl_d2h"] = {"s": round(dt, 3), "rows_per_s": round(n / dt)}
    res["vae_encode_max_abs_diff"] = float(np.abs(mu.cpu().numpy() - ref).max())
    assert res["vae_encode_max_abs_diff"] < 2e-5 * max(1.0, float(np.abs(ref).max()))
print(json.dumps(res, indent=1))
