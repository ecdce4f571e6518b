#!/usr/bin/env python3
"""Per-stage timings on one MI355X (secondary numbers; bench.py is the contract line).
Prints one JSON object; run on the GPU box:  python scripts/bench_stages.py > gpurun_out/stages.json"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from lrbinner_amd import device as lrb, ae_utils, cluster_utils as cu
from bench import synth_packed

dev = torch.device("cuda", 0)
ctx = lrb.Context(0, use_torch_stream=True)
res = {}

def timed(fn, reps=5, warm=1, ramp_ms=60.0):
    # the chip needs ~40 ms of load to reach the clock it then holds (bench.py docstring): warm up by time
    t0 = time.time()
    n_warm = 0
    while n_warm < warm or (time.time() - t0) * 1e3 < ramp_ms:
        fn()
        torch.cuda.synchronize()
        n_warm += 1
        if n_warm >= 200:
            break
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))

# ---- pack: ASCII -> codes+mask+planes -------------------------------------------
n, L = 100_000, 10_000
ascii_t = torch.randint(0, 4, (n * L,), dtype=torch.uint8, device=dev)
ascii_t = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[ascii_t.long()]
offs = (np.arange(n + 1, dtype=np.uint64) * L)
lens, co, mo = lrb.pack_layout(offs)
t = lambda a, dt: torch.from_numpy(a.view(dt)).to(dev)
offs_t, co_t, mo_t = t(offs, np.int64), t(co, np.int64), t(mo, np.int64)
codes = torch.empty(int(co[-1]), dtype=torch.int32, device=dev)
mask = torch.empty(int(mo[-1]), dtype=torch.int32, device=dev)
planes = torch.empty(2 * int(mo[-1]), dtype=torch.int32, device=dev)
vp = lrb.vp
def pack():
    lrb.call("lrb_pack_reads_dev", ctx._h, vp(ascii_t.data_ptr()), n * L, vp(offs_t.data_ptr()), n,
             vp(co_t.data_ptr()), vp(mo_t.data_ptr()), vp(codes.data_ptr()), vp(mask.data_ptr()), vp(planes.data_ptr()))
ms = timed(pack)
res["pack"] = {"ms": ms, "reads_per_s": n / ms * 1e3, "GBps_ascii_in": n * L / ms / 1e6}
del ascii_t

# ---- K1 / K2 / K3 on resident packed reads ----------------------------------------
n = 200_000
codes, mask, co_t, mo_t, lens_t, words = synth_packed(torch, n, L, 1, dev)
pr = lrb.PackedReads(codes, mask, co_t, mo_t, lens_t, n)
ctx.make_planes(pr)
ctx.make_planes_t(pr)
ctx.make_codes_t(pr)
for k, dim in ((3, 32), (4, 136), (5, 512)):
    out = torch.empty((n, dim), dtype=torch.int32, device=dev)
    # the default kernels: lane per read on the group-transposed layouts (planes for k = 3, codes for k = 4, 5)
    fn = (lambda: ctx.kmer_counts3t_dev(pr, out=out)) if k == 3 else (lambda: ctx.kmer_counts4t_dev(pr, out=out, k=k))
    ms = timed(fn)
    alg = (L // 4 + 4 * dim) * n
    res[f"k1_k{k}"] = {"ms": ms, "reads_per_s": n / ms * 1e3, "alg_GBps": alg / ms / 1e6, "hbm_frac": alg / ms / 1e6 / 8000}
    if k != 3:  # the wave-per-read LDS kernel on the per-read layout (round 1's default)
        ms = timed(lambda: ctx.kmer_counts_dev(pr, k, out=out))
        res[f"k1_k{k}_lds"] = {"ms": ms, "reads_per_s": n / ms * 1e3, "hbm_frac": alg / ms / 1e6 / 8000}
    del out
table = torch.zeros(lrb.K15_ENTRIES, dtype=torch.int32, device=dev)
ms = timed(lambda: ctx.k15_accumulate_dev(pr, table), reps=3)
res["k2_accumulate_direct_atomics"] = {"ms": ms, "reads_per_s": n / ms * 1e3, "atomics_per_s": n * (L - 14) / ms * 1e3}
ms = timed(lambda: ctx.k15_accumulate_part_dev(pr, table, n * L), reps=3)
alg = (L // 4 + 8 * (L - 14)) * n
res["k2_accumulate"] = {"ms": ms, "reads_per_s": n / ms * 1e3, "alg_GBps": alg / ms / 1e6, "hbm_frac": alg / ms / 1e6 / 8000,
                        "kmers_per_s": n * (L - 14) / ms * 1e3}
ms = timed(lambda: ctx.k15_mirror_dev(table), reps=3)
res["k2_mirror"] = {"ms": ms, "GBps_rw": 2 * 4 * lrb.K15_ENTRIES / ms / 1e6, "hbm_frac": 2 * 4 * lrb.K15_ENTRIES / ms / 1e6 / 8000}
hist = torch.empty((n, 32), dtype=torch.int32, device=dev); sums = torch.empty(n, dtype=torch.int32, device=dev)
ms = timed(lambda: ctx.cov_hist_dev(pr, table, 10, 32, hist=hist, sums=sums), reps=3)
alg = (L // 4 + 4 * (L - 14) + 4 * 32) * n
res["k3_cov_hist"] = {"ms": ms, "reads_per_s": n / ms * 1e3, "alg_GBps": alg / ms / 1e6, "hbm_frac": alg / ms / 1e6 / 8000,
                      "gathers_per_s": n * (L - 14) / ms * 1e3}
del table, hist, sums, pr, codes, mask

# ---- K4 at Sim-8 scale ---------------------------------------------------------------
N, d, S = 432_333, 4, 1000
rng = np.random.default_rng(0)
centers = rng.normal(size=(8, d)) * 2
lat = (centers[rng.integers(0, 8, N)] + rng.normal(size=(N, d)) * 0.25).astype(np.float32)
be = cu.HipBackend(0)
be.load(lat)
seeds = torch.from_numpy(rng.choice(N, S, replace=False).astype(np.int64)).to(dev)
out = torch.empty((S, 60), dtype=torch.int32, device=dev)
ms = timed(lambda: be.ctx.seed_hist_dev(be.M, seeds, out=out))
res["k4_seed_hist"] = {"ms": ms, "N": N, "dims": d, "seeds": S, "pair_dists_per_s": N * S / ms * 1e3,
                       "unfused_reference_bytes_GB": (1 + S) * (4 * N * d + 4 * N) / 1e9}
dv = torch.empty(N, dtype=torch.float32, device=dev)
ms = timed(lambda: be.ctx.seed_dist_dev(be.M, 5, out=dv))
res["k4_seed_dist"] = {"ms": ms, "GBps": (4 * N * d + 4 * N) / ms / 1e6}
import random
random.seed(1)
t0 = time.time(); clusters = cu.cluster_points(lat, 0, 5000, backend=be); res["cluster_points_exhaustive"] = {"s": time.time() - t0, "N": N, "clusters": len(clusters)}

# ---- VAE ---------------------------------------------------------------------------------
cov = rng.random((N, 10)); comp = rng.dirichlet(np.ones(32), size=N)
vae = ae_utils.VAE(10, 32, latent_dims=4, hidden_layers=[128, 128], device="cuda")
data = ae_utils.make_data(cov, comp, "cuda")
def _train(ne):
    torch.cuda.synchronize(); t0 = time.time()
    vae.trainmodel(data, nepochs=ne, batchsteps=[])
    torch.cuda.synchronize()
    return time.time() - t0
_train(1)
# every trainmodel call builds its trainer and records its graphs: the difference of two run lengths is the epochs alone
t2, t12 = _train(2), _train(12)
per_epoch = (t12 - t2) / 10
res["vae_train"] = {"s_per_epoch": per_epoch, "ms_per_step": per_epoch / (N // 1024) * 1e3, "steps_per_epoch": N // 1024, "batch": 1024,
                    "setup_s_per_call": t2 - 2 * per_epoch}
t0 = time.time(); latv = vae.encode(data); res["vae_encode"] = {"s": time.time() - t0, "rows_per_s": N / (time.time() - t0)}
print(json.dumps(res, indent=1))
