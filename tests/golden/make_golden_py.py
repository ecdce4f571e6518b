#!/usr/bin/env python3
"""Generate VAE / clustering golden vectors by IMPORTING the reference's Python.

Build container only (needs /root/reference).  The reference modules are imported
from where they lie (nothing is copied; PYTHONDONTWRITEBYTECODE is set so no
bytecode is written either); ``Bio`` (absent here) is replaced by a stub whose
``SeqIO.parse`` yields minimal records, which is all cluster_utils needs to write
bins.txt.  The harness seeds ``random`` / ``numpy`` / ``torch`` itself -- the
reference never seeds anything.

Writes tests/golden/py_vae.npz, py_vae_train.npz, py_cluster.npz, py_binning.npz
(inputs + expected outputs only).
"""
import os
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
import pickle
import random
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


class _Rec:
    def __init__(self, rid, seq):
        self.id, self.seq = rid, seq


def _parse(path, fmt):
    rid, parts = None, []
    for line in open(path):
        if line.startswith(">"):
            if rid is not None:
                yield _Rec(rid, "".join(parts))
            rid, parts = line[1:].split()[0], []
        else:
            parts.append(line.strip())
    if rid is not None:
        yield _Rec(rid, "".join(parts))


def import_reference():
    bio = types.ModuleType("Bio")
    seqio = types.ModuleType("Bio.SeqIO")
    seqio.parse = _parse
    bio.SeqIO = seqio
    sys.modules["Bio"] = bio
    sys.modules["Bio.SeqIO"] = seqio
    sys.path.insert(0, REF)
    from mbcclr_utils import ae_utils, cluster_utils
    return ae_utils, cluster_utils


def seed_all(s):
    random.seed(s)
    np.random.seed(s)
    torch.manual_seed(s)


def blobs(rng, n, d, k, spread=0.12):
    centers = rng.normal(size=(k, d))
    lab = rng.integers(0, k, size=n)
    return (centers[lab] + rng.normal(size=(n, d)) * spread).astype(np.float32), lab


def vae_fixture(ae):
    rng = np.random.default_rng(42)
    n, cov_size, prof_size = 1300, 10, 32
    cov = rng.random((n, cov_size)) * rng.random(cov_size)
    cov[:, 7] = 0.0  # constant column, as empty coverage bins are
    comp = rng.dirichlet(np.ones(prof_size), size=n)
    # float32-representable inputs: the fixture stores them as float32 without loss
    cov = cov.astype(np.float32).astype(np.float64)
    comp = comp.astype(np.float32).astype(np.float64)
    seed_all(7)
    vae = ae.VAE(cov_size, prof_size, latent_dims=4, hidden_layers=[32, 32])
    loader = ae.make_data_loader(cov, comp)
    with tempfile.TemporaryDirectory() as tmp:
        vae.trainmodel(loader, nepochs=2, batchsteps=[], save_path=os.path.join(tmp, "m.pt"))
        saved = torch.load(os.path.join(tmp, "m.pt"), weights_only=False)
    enc_loader = ae.make_data_loader(cov, comp, drop_last=False, shuffle=False)
    latent = vae.encode(enc_loader)
    covs_s, profs_s, _ = enc_loader.dataset.tensors
    vae.eval()
    with torch.no_grad():
        mu, logsigma = vae.forward_predict(covs_s[:64], profs_s[:64])
    # loss terms for fixed tensors (no sampling involved)
    g = torch.Generator().manual_seed(3)
    covs_out = torch.rand(64, cov_size, generator=g)
    profs_out = torch.rand(64, prof_size, generator=g)
    loss, e_cov, e_comp, kld = vae.calc_loss(covs_s[:64], covs_out, profs_s[:64], profs_out, mu,
                                             logsigma, torch.arange(64))
    out = {"cov": cov.astype(np.float32), "comp": comp.astype(np.float32), "latent": latent, "covs_scaled": covs_s.numpy(),
           "profs_scaled": profs_s.numpy(), "mu64": mu.numpy(), "logsigma64": logsigma.numpy(),
           "covs_out": covs_out.numpy(), "profs_out": profs_out.numpy(),
           "loss_terms": np.array([loss.item(), e_cov.item(), e_comp.item(), kld.item()]),
           "meta_keys": np.array(sorted(k for k in saved if k != "state")),
           "hidden_layers": np.array(saved["hidden_layers"]),
           "param_count": np.array(ae.count_parameters(vae))}
    for k, v in saved["state"].items():
        out["state." + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "py_vae.npz"), **out)
    print("py_vae.npz:", latent.shape, len(saved["state"]), "tensors")


def vae_train_fixture(ae):
    """The reference's OWN training step (VAE.trainepoch: forward in train mode, calc_loss, backward,
    Adam.step -- ae_utils.py:163-191,199-271) on a fixed 1024-row batch for the C1 (10+32 in, 4 latent)
    and C3 (32+136 in, 8 latent) shapes, two steps each.  The only stochastic inputs are pinned:
    dropoutlayer.p = 0 and torch.randn (the eps of reparameterize) replaced by a known array -- the
    counter-based eps the HIP trainer draws for (seed, step), restated in numpy
    (lrbinner_amd.vae_native.eps_normal) -- so the GPU test can run lrb_vae_train_dev on the same
    batch with the same eps and compare directly.  -> py_vae_train.npz"""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from lrbinner_amd.vae_native import eps_normal
    from torch.utils.data import DataLoader, TensorDataset
    SEED, B = 4242, 1024
    out = {"seed": np.array(SEED), "batch": np.array(B)}
    for tag, cov_size, prof_size, latent in (("c1", 10, 32, 4), ("c3", 32, 136, 8)):
        rng = np.random.default_rng(100 + cov_size)
        X = rng.random((B, cov_size + prof_size)).astype(np.float32)
        X[:, 1] = 0.0   # a constant column, as MinMax scaling of an empty coverage bin gives
        X[:, cov_size:] *= rng.random(prof_size).astype(np.float32)
        seed_all(11 + cov_size)
        vae = ae.VAE(cov_size, prof_size, latent_dims=latent, hidden_layers=[128, 128])
        vae.dropoutlayer.p = 0.0
        out[f"{tag}.X"] = X
        for k, v in vae.state_dict().items():
            out[f"{tag}.init.{k}"] = v.detach().numpy().copy()
        covs, profs = torch.from_numpy(X[:, :cov_size]), torch.from_numpy(X[:, cov_size:])
        loader = DataLoader(TensorDataset(covs, profs, torch.arange(B)), batch_size=B, shuffle=False, drop_last=True)
        optimizer = torch.optim.Adam(vae.parameters(), lr=1e-3)
        real_randn, real_loss, real_step = torch.randn, vae.calc_loss, optimizer.step
        cur = {}

        def fake_randn(*size, **kw):
            assert tuple(size) == (B, latent), size
            return torch.from_numpy(cur["eps"].copy())

        def spy_loss(*a):
            r = real_loss(*a)
            cur["terms"] = np.array([float(x.detach()) for x in r], dtype=np.float64)
            return r

        def spy_step(*a, **kw):
            cur["grads"] = {k: p.grad.detach().numpy().copy() for k, p in vae.named_parameters()}
            return real_step(*a, **kw)

        vae.calc_loss, optimizer.step = spy_loss, spy_step
        for step in range(2):
            cur["eps"] = eps_normal(SEED, step, B, latent)
            torch.randn = fake_randn
            try:
                vae.trainepoch(loader, step, optimizer, set(), None)
            finally:
                torch.randn = real_randn
            out[f"{tag}.s{step}.eps"] = cur["eps"]
            out[f"{tag}.s{step}.loss_terms"] = cur["terms"]       # loss, e_cov, e_comp, kld
            if step == 0:
                for k, g in cur["grads"].items():
                    out[f"{tag}.s0.grad.{k}"] = g
            for k, v in vae.state_dict().items():
                out[f"{tag}.s{step}.post.{k}"] = v.detach().numpy().copy()
        print(tag, "loss terms", out[f"{tag}.s0.loss_terms"], out[f"{tag}.s1.loss_terms"])
    np.savez_compressed(os.path.join(HERE, "py_vae_train.npz"), **out)
    print("py_vae_train.npz:", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "py_vae_train.npz")) // 1024, "KiB")


def cluster_fixture(cu):
    rng = np.random.default_rng(5)
    out = {}
    # normalize, incl. an all-zero row
    m = rng.normal(size=(50, 4)).astype(np.float32)
    m[3] = 0
    out["norm_in"] = m
    out["norm_out"] = cu.normalize(m).numpy()
    # distances + histogram + densities for a few seeds on blobs
    lat, lab = blobs(rng, 6000, 4, 3)
    out["latent"] = lat
    out["labels"] = lab
    M = cu.normalize(lat)
    seeds = np.array([0, 17, 2999, 5999])
    out["seeds"] = seeds
    out["dist"] = np.stack([cu.calc_distances(M, int(s)).numpy() for s in seeds])
    hists = []
    for s in seeds:
        h = torch.histc(cu.calc_distances(M, int(s)), 60, 0, 0.3)
        h[0] -= 1
        hists.append(h.numpy())
    out["hist"] = np.stack(hists)
    out["dens"] = np.stack([cu.calc_densities(torch.from_numpy(h)).numpy() for h in hists])
    # find_valley_ratio: real densities + crafted shapes (SURVEY appendix B probes)
    crafted = [
        [10, 500, 800, 1000, 900, 500, 100, 90, 95] + [0] * 51,
        list(np.linspace(1, 2000, 21)) + [1500, 900, 100] + [0] * 36,   # peak at bin 20: rejected
        list(np.linspace(1, 2000, 20)) + [1500, 900, 100, 50] + [0] * 36,  # peak at bin 19
        [0] * 60,
        [1000, 400, 100, 10, 1] + [0] * 55,                               # peak at x == 0
        list(np.linspace(0, 900, 10)) + list(np.linspace(900, 0, 50)),   # slow descent
        [5, 4, 3, 2, 1, 0, 1, 2, 3] + [0] * 51,
    ]
    for _ in range(12):
        crafted.append(list(np.abs(rng.normal(size=60)).cumsum()[::-1] * rng.integers(1, 400)))
    fv_in = np.array(crafted, dtype=np.float32)
    fv_in = np.concatenate([fv_in, out["dens"]], axis=0)
    fv_out = []
    for row in fv_in:
        r = cu.find_valley_ratio(torch.from_numpy(row))
        fv_out.append([np.nan if (v is False or v is None) else float(v) for v in r])
    out["fv_in"] = fv_in
    out["fv_out"] = np.array(fv_out, dtype=np.float64)
    # get_cluster_center under a fixed python-random stream
    gcc = []
    for seed_pt in (0, 17, 2999):
        random.seed(100 + seed_pt)
        bp, dist, maxima, minima, tail = cu.get_cluster_center(M, seed_pt)
        gcc.append([np.nan if (v is False or v is None) else float(v)
                    for v in (bp, maxima, minima, tail)])
    out["gcc"] = np.array(gcc)
    # cluster_points: exhaustive (iterations == 0) and sampled (iterations > 0)
    for tag, iters in (("exh", 0), ("it", 40)):
        random.seed(11)
        clusters = cu.cluster_points(lat, iters, 500)
        assign = np.full(len(lat), -1, dtype=np.int64)
        for order, (cid, members) in enumerate(clusters.items()):
            assign[np.array(sorted(members), dtype=np.int64)] = order
        out[f"cp_{tag}_assign"] = assign
        out[f"cp_{tag}_n"] = np.array(len(clusters))
    np.savez_compressed(os.path.join(HERE, "py_cluster.npz"), **out)
    print("py_cluster.npz:", {k: np.asarray(v).shape for k, v in out.items() if k.startswith(("cp", "gcc", "fv_out"))})


def binning_fixture(cu):
    """perform_binning end to end on a small synthetic output directory."""
    rng = np.random.default_rng(9)
    n = 4000
    lat, lab = blobs(rng, n, 4, 3)
    lat[-150:] = rng.normal(size=(150, 4)) * 3  # outliers -> left over -> Gaussian assignment
    centers = rng.random((3, 42))
    prof = (centers[lab] + rng.normal(size=(n, 42)) * 0.02).astype(np.float32).astype(np.float64)
    comp, cov = prof[:, :32], prof[:, 32:]
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "profiles"))
        np.save(os.path.join(tmp, "latent.npy"), lat)
        np.save(os.path.join(tmp, "profiles", "com_profs.npy"), comp)
        np.save(os.path.join(tmp, "profiles", "cov_profs.npy"), cov)
        reads = os.path.join(tmp, "reads.fasta")
        lens = rng.integers(5, 60, size=n)
        with open(reads, "w") as f:
            for i in range(n):
                f.write(f">r{i}\n{'A' * int(lens[i])}\n")
        random.seed(21)
        cu.perform_binning(tmp, 0, 300, True, reads)
        bins = np.array([int(x) for x in open(os.path.join(tmp, "bins.txt")).read().split()])
        lengths = np.array([int(x) for x in open(os.path.join(tmp, "lengths.txt")).read().split()])
        res = pickle.load(open(os.path.join(tmp, "binning_result.pkl"), "rb"))
        files = sorted(os.listdir(os.path.join(tmp, "binned_reads")))
        first = open(os.path.join(tmp, "binned_reads", files[0])).read().split("\n")[:2]
    np.savez_compressed(os.path.join(HERE, "py_binning.npz"), latent=lat,
                        comp=comp.astype(np.float32), cov=cov.astype(np.float32),
                        read_lens=lens, bins=bins, lengths=lengths,
                        result_keys=np.array(sorted(res)),
                        result_sizes=np.array([len(res[k]) for k in sorted(res)]),
                        binned_files=np.array(files), first_lines=np.array(first))
    print("py_binning.npz:", np.bincount(bins), files)


def main():
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference")
    ae, cu = import_reference()
    if os.environ.get("GOLDEN_VAE_TRAIN_ONLY"):
        vae_train_fixture(ae)
        return
    if not os.environ.get("GOLDEN_HOST_ONLY"):
        vae_train_fixture(ae)
        vae_fixture(ae)
        cluster_fixture(cu)
        binning_fixture(cu)
    host_fixture()


def host_fixture():
    """Checkpointer transitions and split_contigs of the reference's runners_utils
    (runners_utils.py:16-75) -> py_host.json."""
    import json
    sys.modules.setdefault("metacoag_utils", types.ModuleType("metacoag_utils"))
    sys.modules.setdefault("metacoag_utils.marker_gene_utils", types.ModuleType("marker_gene_utils"))
    sys.modules["metacoag_utils"].marker_gene_utils = sys.modules["metacoag_utils.marker_gene_utils"]
    from mbcclr_utils import runners_utils as R
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        cp = R.Checkpointer(os.path.join(tmp, "ck"))
        script = [("run?", "1_1", ["r.fa", 3]), ("log", "1_1", ["r.fa", 3]), ("log", "1_2", ["r.fa"]),
                  ("log", "2_1", ["r.fa", 10, 32]), ("log", "3_1", ["numpy"]), ("log", "4_1", ["o", 8, [128, 128], 200, None]),
                  ("run?", "1_1", ["r.fa", 3]), ("run?", "1_1", ["r.fa", 4]), ("log", "2_1", ["r.fa", 5, 32]),
                  ("run?", "3_1", ["numpy"]), ("run?", "1_2", ["r.fa"]), ("log", "1_1", ["r.fa", 4]), ("run?", "2_1", ["r.fa", 5, 32])]
        trace = []
        for op, stage, params in script:
            if op == "run?":
                trace.append(["run?", stage, params, bool(cp.should_run_step(stage, params))])
            else:
                cp.log(stage, params)
                trace.append(["log", stage, params, sorted(cp.completed)])
        cp2 = R.Checkpointer(os.path.join(tmp, "ck"), True)
        out["checkpoint_trace"] = trace
        out["checkpoint_reload"] = sorted(cp2.completed)
        # split_contigs
        rng = np.random.default_rng(4)
        lens = [100, 4999, 5000, 5001, 7500, 12345, 2500]
        os.makedirs(os.path.join(tmp, "fragments"))
        fa = os.path.join(tmp, "contigs.fasta")
        with open(fa, "w") as f:
            for i, L in enumerate(lens):
                s = "".join(rng.choice(list("ACGT"), size=L))
                f.write(f">contig_{i} some description\n")
                for j in range(0, L, 60):
                    f.write(s[j:j + 60] + "\n")
        groups, parent = R.split_contigs(fa, tmp)
        out["contig_lens"] = lens
        out["contigs_fasta"] = open(fa).read()
        out["fragments_fasta_sha"] = __import__("hashlib").sha256(open(os.path.join(tmp, "fragments", "contigs.fasta"), "rb").read()).hexdigest()
        out["groups"] = {k: v for k, v in groups.items()}
        out["parent"] = {str(k): v for k, v in parent.items()}
    with open(os.path.join(HERE, "py_host.json"), "w") as f:
        json.dump(out, f)
    print("py_host.json:", len(out["checkpoint_trace"]), "checkpoint steps,", len(out["parent"]), "fragments")


if __name__ == "__main__":
    main()
