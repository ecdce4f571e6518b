#!/usr/bin/env python3
"""bench.py -- reads/s of the composition kernel (K1) on synthetic 10 kb reads.

Workload (BASELINE.json configs[1]): 1 M synthetic 10 kb reads per GPU, k=3,
canonical k-mer tallies only, packed reads already resident in HBM when the
timed region starts.  A "step" is one pass of K1 over the whole batch.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (contract in the task statement; < 8 KB) carrying
  roofline     -- algorithmic bytes (ceil(L/4) + 4*dim per read) / mean K1 launch
                  duration measured with HIP events on the launch stream, vs 8 TB/s; and INSIDE it (a record that keeps
                  `roofline` keeps every fraction):
                    roofline.stages[name] = {frac, kernel_ms, traffic_ratio, bound} -- the other kernels of the path
                      against their rooflines: K1 k=4/5 (warm: 30 + 100 launches), K2, K3 by BOTH routes (`k3_default`: the
                      windows partitioned again; `k3_kept_lists`: the sweep alone -- what the product does for the last group
                      of its table stage, and for every group with LRB_KEEP_LISTS=1), K3 at 64 bins, K4, K5, encode, K6,
                      the VAE training step by batch size
                    roofline.c4_rank -- the path that HAS the collective (BASELINE configs[3] shape): per rank
                      2.5 M reads, timed through the product's own objects (lrbinner_amd.dist.HipCompute, the calls of
                      profile_file_sharded): K1 k=4 (one launch over all resident batches) -> K2 into the canonical half ->
                      all-reduce of it (RCCL) -> expand -> K3; reads/s by route, per-phase ms as maxima over the ranks and
                      per rank, the world RCCL saw, all-reduce ms / bytes / bus GB/s.  Not part of `value`.
  cpu_baseline -- the reference's own count-kmers binary (oracle/_ref, kind
                  "reference") or the oracle port, timed on this box's host cores
                  on a bounded sample of the same reads (N=1 only)
Everything else a run measures (per-kernel traffic, both C4 routes in full, cold figures, the CPU legs of the 15-mer
executables) goes to the detail file: LRB_BENCH_DETAIL, default gpurun_out/bench_detail.json.
Reads shard across ranks with no data-path collective for K1 (weak scaling:
every rank owns 1 M reads); the collective lives in `roofline.c4_rank`.

Clocks: from idle the chip needs ~40 ms of sustained load to reach the clock it then
holds (measured: the 0.80 ms launch of the first steps settles at 0.73 ms; with 5 timed
steps after 2 / 10 / 50 warm-up steps the rate is 1.11 / 1.24 / 1.33 G reads/s).  The
set-up therefore ends with --clock-ramp-ms (100 ms) of the same kernel, untimed and outside
the W warm-up steps, so that a short run measures the state a long one is in; the timed
region is exactly K steps as the contract says.  Defaults: K = 200, W = 50 (0.2 s).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def synth_packed(torch, n, L, seed, device):
    """n uniform-random reads of L bases directly in the packed HBM layout.
    Random 32-bit words ARE uniform 2-bit bases; every base is valid ACGT."""
    ncw = -(-L // 16)
    words = ((ncw + 3) // 4) * 4 + 4
    nmw = -(-L // 32)
    mwords = ((nmw + 3) // 4) * 4 + 4
    g = torch.Generator(device=device).manual_seed(seed)
    codes = torch.randint(-2 ** 31, 2 ** 31 - 1, (n, words), dtype=torch.int32, device=device,
                          generator=g)
    codes[:, ncw:] = 0
    if L % 16:
        keep = (-1 << (2 * (16 - L % 16))) & 0xFFFFFFFF
        codes[:, ncw - 1] &= np.int32(np.uint32(keep).view(np.int32))
    mask = torch.zeros((n, mwords), dtype=torch.int32, device=device)
    mask[:, :nmw] = -1
    if L % 32:
        keep = (-1 << (32 - L % 32)) & 0xFFFFFFFF
        mask[:, nmw - 1] = int(np.uint32(keep).view(np.int32))
    co = torch.arange(n + 1, dtype=torch.int64, device=device) * words
    mo = torch.arange(n + 1, dtype=torch.int64, device=device) * mwords
    lens = torch.full((n,), L, dtype=torch.int32, device=device)
    return codes.view(-1), mask.view(-1), co, mo, lens, words


def unpack_to_fasta(codes_host, words, n, L, path):
    """Write the first n packed reads as FASTA (same reads the GPU processed)."""
    letters = np.frombuffer(b"ACTG", dtype=np.uint8)
    i = np.arange(L)
    sh = (30 - 2 * (i % 16)).astype(np.uint32)
    with open(path, "wb") as f:
        for r in range(n):
            w = codes_host[r * words:(r + 1) * words]
            seq = letters[(w[i // 16] >> sh) & 3]
            f.write(b">r%d\n" % r)
            f.write(seq.tobytes())
            f.write(b"\n")


def cpu_baseline(codes_host, words, L, k, sample, sample15=(20_000, 80_000)):
    """The reference's own executables (oracle/_ref: the reference sources compiled by plain g++) on this box's host
    cores, on a bounded sample of the same synthetic reads written as FASTA to tmpfs:
      count-kmers     (count-kmers.cpp:66-190)     `sample` reads, file -> com_profs text
      count-15mers    (count-15mers.cpp:97-123)    two sample sizes: the fixed cost (zero + dump the 4 GiB table) and the
                                                   marginal reads/s come apart from the two times
      search-15mers   (search-15mers.cpp:121-157)  the same two samples against the larger sample's table (bs 10, bc 32)
    and, on the SAME FASTA the CPU leg read, this build's run_kmers file -> com_profs on the GPU (same boundary on both
    sides: that ratio is `gpu_over_cpu_file_to_file`).  Without the reference binaries: the scalar C port, counts only."""
    from oracle import oracle as orc
    cores = os.cpu_count() or 1
    scratch = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=scratch) as tmp:
        fa = os.path.join(tmp, "sample.fasta")
        unpack_to_fasta(codes_host, words, sample, L, fa)
        ref = orc.ref_bin("count-kmers")
        if not ref:
            buf, offs = orc.fastx_read(fa)
            sub = min(sample, 20000)
            t0 = time.perf_counter()
            orc.count_kmers(buf, offs[: sub + 1], k)
            dt = time.perf_counter() - t0
            return {"value": sub / dt, "unit": "reads/s", "cores": 1, "kind": "port",
                    "sample": f"{sub} reads, scalar C oracle (counts only, no text)", "seconds": dt}

        def run(cmd):
            t0 = time.perf_counter()
            subprocess.run([str(c) for c in cmd], check=True, stdout=subprocess.DEVNULL)
            return time.perf_counter() - t0

        dt = run([ref, fa, os.path.join(tmp, "com_profs"), k, cores])
        res = {"value": sample / dt, "unit": "reads/s", "cores": cores, "kind": "reference",
               "sample": f"{sample} of the same synthetic {L}-base reads as FASTA in tmpfs, "
                         f"count-kmers k={k} threads={cores}, wall incl. file read + text write",
               "seconds": dt}
        os.remove(os.path.join(tmp, "com_profs"))
        # the same boundary on the GPU side: this build's run_kmers, file -> com_profs (second of two runs: the
        # first one pays the context and the workspaces)
        try:
            from lrbinner_amd import runners_utils as ru
            og = os.path.join(tmp, "gpu")
            ru.run_kmers(fa, og, k, 32)
            t0 = time.perf_counter()
            ru.run_kmers(fa, og, k, 32)
            dg = time.perf_counter() - t0
            res["gpu_file_to_file"] = {"reads_per_s": sample / dg, "seconds": dg,
                                       "what": f"lrbinner_amd.runners_utils.run_kmers on the same FASTA -> com_profs text (k={k})"}
            res["gpu_over_cpu_file_to_file"] = dt / dg
            import shutil
            shutil.rmtree(og, ignore_errors=True)
        except Exception as e:  # noqa: BLE001
            res["gpu_file_to_file"] = {"error": f"{type(e).__name__}: {e}"}
        c15, s15 = orc.ref_bin("count-15mers"), orc.ref_bin("search-15mers")
        if c15 and s15 and sample15 and sample15[1] <= sample:
            s1, s2 = sample15
            fa1, fa2 = os.path.join(tmp, "s1.fasta"), os.path.join(tmp, "s2.fasta")
            unpack_to_fasta(codes_host, words, s1, L, fa1)
            unpack_to_fasta(codes_host, words, s2, L, fa2)
            t1p, t2p = os.path.join(tmp, "t1"), os.path.join(tmp, "t2")
            a1 = run([c15, fa1, t1p, cores])
            os.remove(t1p)
            a2 = run([c15, fa2, t2p, cores])

            def split(x1, x2):   # time = fixed + reads / rate
                rate = (s2 - s1) / (x2 - x1) if x2 > x1 else float("inf")
                return {"marginal_reads_per_s": rate, "fixed_s": max(0.0, x1 - s1 / rate), "seconds": [x1, x2],
                        "gross_reads_per_s": s2 / x2}

            res["count_15mers"] = dict(split(a1, a2), cores=cores, kind="reference",
                                       sample=f"{s1} and {s2} reads, count-15mers threads={cores}, incl. the 4 GiB table file "
                                              "written to tmpfs (the fixed part)")
            b1 = run([s15, t2p, fa1, os.path.join(tmp, "cov1"), 10, 32, cores])
            b2 = run([s15, t2p, fa2, os.path.join(tmp, "cov2"), 10, 32, cores])
            res["search_15mers"] = dict(split(b1, b2), cores=cores, kind="reference",
                                        sample=f"{s1} and {s2} reads against the {s2}-read table, search-15mers bs=10 bc=32 "
                                               f"threads={cores}, incl. loading the 4 GiB table (the fixed part)")
        return res


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks here, one per GPU,
    as a CHILD job (`python -m torch.distributed.run`, rendezvous on 127.0.0.1) and leave with its exit code.
    Called before torch is imported or the GPU is touched; the child is a new process, never an exec of this
    one.  Under a launcher (RANK in the environment) the world the launcher made has to be the one asked for."""
    if "RANK" in os.environ:
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world != args.gpus and not (world == 1 and args.gpus <= 1):
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks\n")
            sys.exit(2)
        return
    if args.gpus <= 1:
        return
    # N devices have to BE there: a rendezvous of N ranks on fewer GPUs does not fail, it hangs (RCCL refuses two
    # ranks on a device only after the group is up).  Counted from sysfs -- nothing here touches a GPU.
    from lrbinner_amd import _gpus
    have = _gpus.visible_gpus()
    if have is not None and have < args.gpus and os.environ.get("LRB_BENCH_BACKEND", "nccl") == "nccl":
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but this node shows {have} GPU(s)\n")
        sys.exit(2)
    # torchrun picks the rendezvous port itself (--standalone: a c10d store on a free port of 127.0.0.1) -- a port
    # chosen here by bind-then-close could be taken before the child binds it
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--standalone", "--local-addr", "127.0.0.1", "--master-addr", "127.0.0.1",
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    sys.exit(subprocess.run(cmd, env=env).returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--clock-ramp-ms", type=float, default=100.0,
                    help="untimed GPU work before the W warm-up steps (see the module docstring)")
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads per GPU")
    ap.add_argument("--read-len", type=int, default=10_000)
    ap.add_argument("--k", type=int, default=3)
    ap.add_argument("--cpu-sample", type=int, default=300_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--no-c4", action="store_true", help="skip the C4-shaped phase set (c4_phases)")
    ap.add_argument("--c4-reads", type=int, default=2_500_000, help="reads per GPU of the C4-shaped phase set")
    ap.add_argument("--force-collective", action="store_true",
                    help="c4_phases: run fold -> all-reduce -> expand even with one rank (exercises the N > 1 code on one GPU)")
    ap.add_argument("--stages-child", action="store_true", help=argparse.SUPPRESS)  # rocprofv3 child of roofline_stages
    ap.add_argument("--no-traffic", action="store_true",
                    help="do not measure roofline.traffic with rocprofv3 child runs (N=1 only)")
    ap.add_argument("--k1-mode", type=int, default=0,
                    help="0 the library default: lane-per-read kernels on the group-transposed layouts (bit planes "
                         "for k=3, codes for k=4,5); 1 wave-per-read LDS-histogram kernel on the per-read codes; "
                         "4 (k=3) 4-mers at even positions on the codes layout (k1_lane4s2_kernel<.., 3>)")
    args = ap.parse_args()
    launch_ranks(args)  # --gpus N > 1 without a launcher: N ranks as a child job; never returns in that case

    import torch
    import torch.distributed as dist
    from lrbinner_amd import device as lrb

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or "RANK" in os.environ  # launched by torch.distributed.run
    # rehearsal hook: LRB_BENCH_BACKEND=gloo runs several ranks on ONE GPU (RCCL refuses two ranks on a device) to
    # exercise the multi-rank code path where only a single MI355X is at hand; the numbers of such a run mean nothing
    backend = os.environ.get("LRB_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        from lrbinner_amd import dist as ld
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=ld.collective_timeout())
        else:
            dist.init_process_group(backend, timeout=ld.collective_timeout())
    numa = None
    if world > 1:
        from lrbinner_amd import _gpus
        numa = _gpus.pin_to_gpu_numa(local)     # the rank's host-side work next to its GPU
    if os.environ.get("LRB_BENCH_FAIL_RANK") == str(rank) and world > 1:
        sys.exit(3)   # tests: a rank that dies before the first collective (tests/test_gpu_multi.py)
    dev = torch.device("cuda", local)
    n, L, k = args.reads, args.read_len, args.k
    dim = lrb.kmer_dim(k)

    ctx = lrb.Context(local, use_torch_stream=True)
    codes, mask, co, mo, lens, words = synth_packed(torch, n, L, 12345 + rank, dev)
    pr = lrb.PackedReads(codes, mask, co, mo, lens, n)
    if args.stages_child:   # a few launches of the other kernels for the PMC passes of roofline_stages, nothing else
        roofline_stages(torch, lrb, ctx, pr, dev, L, reps=6, traffic=False, bins64=False)
        ctx.close()
        return
    out = torch.empty((n, dim), dtype=torch.int32, device=dev)

    # layouts of the resident reads (outside the timed region, like packing itself)
    if k == 3 and args.k1_mode == 4:
        ctx.make_codes_t(pr, sort=True)
    elif k == 3:
        ctx.make_planes(pr)
        if args.k1_mode == 0:
            ctx.make_planes_t(pr, sort=True)
    elif args.k1_mode == 0:
        ctx.make_codes_t(pr, sort=True)

    def step():
        if k == 3 and args.k1_mode == 4:
            ctx.kmer_counts4t_dev(pr, out=out, k=3)     # 4-mers at even positions (k1_lane4s2_kernel<.., 3>)
        elif k == 3 and args.k1_mode == 0:
            ctx.kmer_counts3t_dev(pr, out=out)
        elif k == 3:
            ctx.kmer_counts3_dev(pr, mode=args.k1_mode, out=out)
        elif args.k1_mode == 0:
            ctx.kmer_counts4t_dev(pr, out=out, k=k)      # lane per read on group-transposed codes
        else:
            ctx.kmer_counts_dev(pr, k, out=out)          # wave per read on the per-read layout

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # cold figure: the first W + K launches from an idle GPU, no ramp (reported as roofline.frac_cold
    # next to the steady-state frac; the chip reaches its load clock ~40 ms into a burst)
    torch.cuda.synchronize()
    time.sleep(0.3)
    n_cold = args.warmup + args.steps
    evc = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_cold)]
    for a, b in evc:
        a.record()
        step()
        b.record()
    torch.cuda.synchronize()
    cold_all = [a.elapsed_time(b) for a, b in evc]
    cold_ms = float(np.mean(cold_all))
    cold_first_ms = float(np.mean(cold_all[: min(20, n_cold)]))

    # clock ramp (set-up, not part of W or K): the same kernel until the chip holds its load clock
    if args.clock_ramp_ms > 0:
        torch.cuda.synchronize()
        t_ramp = time.perf_counter()
        while (time.perf_counter() - t_ramp) * 1e3 < args.clock_ramp_ms:
            for _ in range(8):
                step()
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    fence()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        step()
        b.record()
    fence()
    dt = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # sanity: row sums must be L-k+1 (cheap, outside the timed region)
    assert int(out[:1024].sum(dim=1).min().item()) == L - k + 1

    alg_bytes = (-(-L // 4) + 4 * dim) * n
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    kernel_name = ("k1_swar3_lane_kernel" if args.k1_mode == 0 else "k1_count_kernel<3>") if k == 3 \
        else (f"k1_lane4_kernel<{k}>" if args.k1_mode == 0 else f"k1_count_kernel<{k}>")
    # HBM bytes per launch: measured in this run by two rocprofv3 PMC child runs of this script
    # (FETCH_SIZE, WRITE_SIZE; separate passes, gfx950 correction of MI355X_MICROARCH.md) when the
    # profiler is there, else the committed figure of the same kernel and workload shape
    traffic, traffic_src = None, None
    if rank == 0 and world == 1 and not args.no_traffic and not os.environ.get("LRB_BENCH_CHILD"):
        try:
            traffic = measure_traffic(kernel_name.split("<")[0], n, L, k, args.k1_mode)
            traffic_src = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child runs of this command (2 x FETCH + WRITE)"
        except Exception as e:  # noqa: BLE001
            traffic_src = f"in-run measurement failed ({type(e).__name__}: {e})"
    if traffic is None:
        try:
            with open(os.path.join(ROOT, "profiles", "k1_traffic.json")) as f:
                tj = json.load(f)
            if tj["kernel"] == kernel_name and tj["workload"]["read_len"] == L and tj["workload"]["k"] == k:
                traffic = tj["bytes_per_read"] * n
                traffic_src = "profiles/k1_traffic.json" + (f" ({traffic_src})" if traffic_src else "")
        except (OSError, KeyError, ValueError):
            pass
    line = {
        "metric": "long reads binned/sec (k=3, 10 kb reads): composition-vector stage",
        "value": n * world * args.steps / dt,
        "unit": "reads/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {"workload": f"{n} synthetic {L}-base reads per GPU, k={k} canonical k-mer "
                               f"tallies (K1) only, packed reads resident in HBM "
                               f"(BASELINE configs[1])",
                   "reads_per_gpu": n, "read_len": L, "k": k, "dim": dim, "clock_ramp_ms": args.clock_ramp_ms,
                   "sharding": "reads split by rank, no collective"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_unit": "bytes per launch (rocprofv3 FETCH_SIZE*2 + WRITE_SIZE)",
                     "traffic_source": traffic_src,
                     "frac_cold": alg_bytes / (cold_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "kernel_ms_cold": cold_ms,
                     "frac_cold_first20": alg_bytes / (cold_first_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "cold_definition": f"mean over the first {n_cold} launches (W + K) from an idle GPU, no clock ramp; "
                                        "first20 = its first 20 launches",
                     "kernel": kernel_name, "kernel_ms": kern_ms,
                     "algorithmic_bytes_per_read": -(-L // 4) + 4 * dim},
    }

    if numa is not None:
        line["config"]["numa_pin_rank0"] = numa
    if k == 3 and args.k1_mode == 0:
        # what actually bounds this kernel: vector-ALU issue, 82 instructions per
        # 32-base block per lane-read (ISA + SQ_INSTS_VALU), one wave64 instruction per clock per CU;
        # priced at the 2.4 GHz maximum clock (MI355X_MICROARCH.md) -- informational
        wave_instr = (n / 64.0) * (-(-L // 32)) * 82
        issue_s = wave_instr / (256 * 2.4e9)
        line["roofline"]["issue_bound"] = {"valu_instr_per_32_bases": 82, "cus": 256, "clock_ghz": 2.4,
                                           "min_kernel_ms": issue_s * 1e3, "frac_of_issue_peak": issue_s / (kern_ms * 1e-3)}
    host_sample = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sample = min(args.cpu_sample, n)
        host_sample = codes[: sample * words].cpu().numpy().view(np.uint32)
    if not args.no_extra:
        try:  # secondary numbers must never cost the contract line
            line["extra"] = extra_stages(torch, dist, lrb, ctx, pr, use_dist, dev, min(n, 100_000), L)
            ok = 1
        except Exception as e:  # noqa: BLE001
            line["extra"] = {"error": f"{type(e).__name__}: {e}"}
            ok = 0
        if use_dist:  # a rank that failed inside a collective would leave the others waiting: fail together
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)

    if not args.no_extra:
        try:  # the other kernels of the path against their own rooflines (SURVEY 8(d) bytes per read)
            line["roofline_stages"] = roofline_stages(
                torch, lrb, ctx, pr, dev, L, reps=10,
                traffic=rank == 0 and world == 1 and not args.no_traffic and not os.environ.get("LRB_BENCH_CHILD"))
            for kk in (4, 5):   # (the keys earlier rounds quoted)
                st = line["roofline_stages"][f"k1_k{kk}"]
                line["extra"].update({f"k1_k{kk}_ms": st["kernel_ms"], f"k1_k{kk}_reads": n,
                                      f"k1_k{kk}_reads_per_s": n / (st["kernel_ms"] * 1e-3),
                                      f"k1_k{kk}_roofline_frac": st["frac"], f"k1_k{kk}_kernel": st["kernel"]})
        except Exception as e:  # noqa: BLE001
            line["roofline_stages"] = {"error": f"{type(e).__name__}: {e}"}
    if not args.no_extra and rank == 0:
        try:  # K7, the kernel every whole run waits for: the fused VAE training step, per batch size of the schedule
            line["vae_step"] = vae_step_times(torch, lrb)
        except Exception as e:  # noqa: BLE001
            line["vae_step"] = {"error": f"{type(e).__name__}: {e}"}
    if not args.no_c4:
        del out
        pr.planes = pr.planes_t = None
        del pr, codes, mask
        torch.cuda.empty_cache()
        try:
            line["c4_phases"] = c4_phases(torch, dist, lrb, use_dist, dev, rank, world, local, args.c4_reads, L,
                                          force_collective=args.force_collective and use_dist)
            ok = 1
        except Exception as e:  # noqa: BLE001
            line["c4_phases"] = {"error": f"{type(e).__name__}: {e}"}
            ok = 0
        if use_dist:
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sample = min(args.cpu_sample, n)
        ctx.sync()
        line["cpu_baseline"] = cpu_baseline(host_sample, words, L, k, sample)
        # NOT the same boundary on both sides: the GPU figure is the kernel on reads resident in HBM, the CPU figure a
        # file -> file executable (FASTA parse + "%f" text included); the like-for-like ratio is gpu_over_cpu_file_to_file
        line["cpu_baseline"]["gpu_kernel_only_over_cpu_file_to_file"] = line["value"] / line["cpu_baseline"]["value"]
        rs = line.get("roofline_stages") or {}
        for name, stage in (("count_15mers", "k2"), ("search_15mers", "k3_default")):
            cb = line["cpu_baseline"].get(name)
            if cb and stage in rs and "kernel_ms" in rs[stage]:
                cb["gpu_kernels_reads_per_s"] = rs[stage]["reads"] / (rs[stage]["kernel_ms"] * 1e-3)
                cb["gpu_kernels_over_cpu_marginal"] = cb["gpu_kernels_reads_per_s"] / cb["marginal_reads_per_s"]
    elif rank == 0:
        line["cpu_baseline"] = None

    if rank == 0:
        print(json.dumps(compact_line(line)), flush=True)
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


def compact_line(line):
    """The contract line (< 8 KB, so that a record that keeps its tail keeps all of it) with every stage's figures INSIDE
    `roofline` -- the key a driver keeps:
      roofline.stages[name]  = {frac, kernel_ms, traffic_ratio (measured HBM bytes / algorithmic bytes, or null), bound}
      roofline.c4_rank       = the C4-shaped rank: reads/s by route, the collective's time / bytes / bus GB/s,
                               the world RCCL saw, per-rank phase times beside the maxima
    Everything else the run measured (per-kernel traffic, notes, cold figures of the stages, the VAE step by batch size,
    the CPU legs of the 15-mer executables) goes to the detail file (LRB_BENCH_DETAIL, default gpurun_out/bench_detail.json;
    copied to profiles/ per round): definitions live in DESIGN.md 6, not in the line."""
    detail = {k_: line.get(k_) for k_ in ("extra", "roofline_stages", "vae_step", "c4_phases", "cpu_baseline", "roofline", "config")}
    out = {k_: v for k_, v in line.items() if k_ not in ("extra", "roofline_stages", "vae_step", "c4_phases")}
    rf = dict(out["roofline"])
    for k_ in ("traffic_unit", "traffic_source", "cold_definition"):
        rf.pop(k_, None)
    r3 = lambda x: None if x is None else float(f"{x:.4g}")
    stages = {}
    rs = line.get("roofline_stages") or {}
    for name, st in rs.items():
        if not isinstance(st, dict) or "kernel_ms" not in st:
            continue
        alg = None
        if st.get("algorithmic_bytes_per_read") and st.get("reads"):
            alg = st["algorithmic_bytes_per_read"] * st["reads"]
        elif st.get("algorithmic_bytes"):
            alg = st["algorithmic_bytes"]
        ratio = st["traffic"] / alg if st.get("traffic") and alg else None
        stages[name] = {"frac": r3(st.get("frac")), "kernel_ms": r3(st["kernel_ms"]), "traffic_ratio": r3(ratio), "bound": st.get("bound")}
    b64 = rs.get("k3_bins64") or {}
    for nm in ("default", "kept_lists"):
        if isinstance(b64.get(nm), dict):
            stages[f"k3_bins64_{nm}"] = {"frac": r3(b64[nm]["frac"]), "kernel_ms": r3(b64[nm]["kernel_ms"]), "traffic_ratio": None, "bound": "hbm"}
    vs = line.get("vae_step") or {}
    for shape in ("c1_shape", "c3_shape"):
        for bs, e in (vs.get(shape) or {}).items():
            stages[f"vae_step_{shape[:2]}_b{bs}"] = {"frac": r3(e["mfma_frac"]), "kernel_ms": r3(e["us"] * 1e-3), "traffic_ratio": None, "bound": "mfma"}
    if "error" in rs:
        stages["error"] = rs["error"]
    rf["stages"] = stages
    c4 = line.get("c4_phases") or {}
    if "routes" in c4 and "error" not in c4:
        d, kp = c4["routes"]["default"], c4["routes"].get("kept_lists", {})
        ph = d["phases_ms_max_over_ranks"]
        rf["c4_rank"] = {"reads_per_gpu": c4["reads_per_gpu"], "world_size_seen_by_rccl": c4["world_size_seen_by_rccl"],
                         "default_reads_per_s": r3(d["reads_per_s"]), "kept_reads_per_s": r3(kp.get("reads_per_s")),
                         "with_text_reads_per_s": r3(d["with_text_reads_per_s"]),
                         "with_text_serial_reads_per_s": r3(d.get("with_text_serial_reads_per_s")),
                         "kept_with_text_reads_per_s": r3(kp.get("with_text_reads_per_s")),
                         "phases_ms_max_over_ranks": {k_: r3(v) for k_, v in ph.items()},
                         "with_text_phases_ms": {k_: r3(v) for k_, v in d["with_text_ms"].items()},
                         "phases_ms_per_rank": d.get("phases_ms_per_rank"),
                         "allreduce": c4.get("allreduce"), "allreduce_ms": r3(c4.get("allreduce_ms")), "allreduce_bytes": c4.get("allreduce_bytes"),
                         "allreduce_busbw_GBps": r3(c4.get("allreduce_busbw_GBps")), "collective_via": c4.get("collective_via"),
                         "allreduce_model_ms": {k_: r3(v) for k_, v in (c4.get("allreduce_cost_model") or {}).items() if k_.endswith("_ms")}}
        if "error" in kp:
            rf["c4_rank"]["kept_error"] = str(kp["error"])[:160]
    elif c4:
        rf["c4_rank"] = {"error": c4.get("error")}
    out["roofline"] = rf
    cb = out.get("cpu_baseline")
    if cb:
        keep = ("value", "unit", "cores", "kind", "sample", "seconds", "gpu_over_cpu_file_to_file", "gpu_kernel_only_over_cpu_file_to_file")
        slim = {k_: cb[k_] for k_ in keep if k_ in cb}
        slim["sample"] = str(slim.get("sample", ""))[:110]
        for leg in ("count_15mers", "search_15mers"):
            if isinstance(cb.get(leg), dict):
                slim[leg] = {k_: r3(cb[leg].get(k_)) for k_ in ("marginal_reads_per_s", "fixed_s", "gpu_kernels_over_cpu_marginal")}
        out["cpu_baseline"] = slim
    path = os.environ.get("LRB_BENCH_DETAIL", os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump({"line": out, "detail": detail}, f, indent=1)
        out["detail_file"] = os.path.relpath(path, ROOT)
    except OSError:
        out["detail_file"] = None
    return out


def collect_counters(child_args, counters=("FETCH_SIZE", "WRITE_SIZE"), timeout=240):
    """{counter: {kernel name up to '(': mean value}} from one rocprofv3 --pmc child run of this script per counter
    (separate passes, as MI355X_MICROARCH.md prescribes; the program stands directly behind `--`)."""
    import csv
    import shutil
    prof = shutil.which("rocprofv3")
    if not prof:
        raise RuntimeError("rocprofv3 not on PATH")
    res = {}
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        for counter in counters:
            d = os.path.join(tmp, counter)
            cmd = [prof, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "k1", "--", sys.executable,
                   os.path.abspath(__file__)] + child_args
            subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout,
                           env=dict(os.environ, LRB_BENCH_CHILD="1", TMPDIR="/tmp"), cwd="/tmp")
            got = {}
            for root, _, files in os.walk(d):
                for fn in files:
                    if fn.endswith("counter_collection.csv"):
                        for r in csv.DictReader(open(os.path.join(root, fn))):
                            if r["Counter_Name"] == counter:
                                got.setdefault(r["Kernel_Name"].split("(")[0], []).append(float(r["Counter_Value"]))
            res[counter] = {k_: float(np.mean(v)) for k_, v in got.items()}
    return res


def vae_step_times(torch, lrb, rows=200_000):
    """K7 (csrc/lrb_vae.hip): microseconds per optimiser step of the fused VAE training step at the batch sizes of the
    reference's schedule (ae_utils.py:199-241: 1024 rows, doubled at epochs 50 / 100 / 150), for the network of C1 / C2
    (10 + 32 -> 128-128 -> 4) and of C3 / C4 / C5 (32 + 136 -> 128-128 -> 8), on synthetic rows; `mfma_frac` = the
    step's fp32 multiply-adds (forward, dX and dW: 3 x 2 x rows x weights) against the dense fp32 MFMA peak
    (256 CUs x 256 flop / clock x 2.4 GHz = 157.3 TFLOP/s) -- the step is a chain of launch latencies, not MFMA-bound."""
    from lrbinner_amd import ae_utils
    from lrbinner_amd.vae_native import NativeTrainer
    res = {"unit": "us per step", "launches_per_step": 11, "mfma_peak_TFLOPs_fp32": 157.3}
    for name, cov, prof, latent in (("c1_shape", 10, 32, 4), ("c3_shape", 32, 136, 8)):
        data = torch.rand(rows, cov + prof, device="cuda")
        perm = torch.randperm(rows, device="cuda")
        vae = ae_utils.VAE(cov, prof, latent_dims=latent, hidden_layers=[128, 128], device="cuda")
        w = ae_utils.h_params[str(prof)]
        vctx = lrb.Context(0, use_torch_stream=True)
        tr = NativeTrainer(vctx, vae, 8192, [w["e_cov_weight"], w["e_comp_weight"], w["kld_weight"]])
        try:
            tr.push()
            d = cov + prof
            weights = d * 128 + 128 * 128 + 128 * 2 * latent + latent * 128 + 128 * 128 + 128 * d
            ent = {}
            for bs in (1024, 2048, 4096, 8192):
                nb = rows // bs
                tr.train(data, perm, bs, nb)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    tr.train(data, perm, bs, nb)
                torch.cuda.synchronize()
                us = (time.perf_counter() - t0) / (3 * nb) * 1e6
                ent[str(bs)] = {"us": us, "mfma_frac": 6.0 * bs * weights / (us * 1e-6) / 157.3e12}
            res[name] = ent
        finally:
            tr.close()
            vctx.close()
    return res


def roofline_stages(torch, lrb, ctx, pr, dev, L, reps=10, traffic=True, bins64=True):
    """The kernels of the path other than the headline one, each against the HBM roofline with SURVEY 8(d)'s
    algorithmic bytes per read: K1 at k = 4 and k = 5 (lane-per-read kernels, all resident reads), K2 (slice lists
    -> canonical half: part + split + tally kernels) and K3 (the sweep of the kept lists) on 400 k reads = 4e9
    windows.  kernel_ms from HIP events on the launch stream; traffic = HBM bytes per launch (2 x FETCH_SIZE +
    WRITE_SIZE of rocprofv3 --pmc child runs of this script, per kernel, summed per stage)."""
    n = pr.n
    res = {}

    def timed(fn, reps_):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps_):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps_

    def entry(kernels, ms, bytes_per_read, reads):
        ach = bytes_per_read * reads / (ms * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": kernels if isinstance(kernels, str) else None,
                "kernels": None if isinstance(kernels, str) else kernels, "kernel_ms": ms, "reads": reads,
                "algorithmic_bytes_per_read": bytes_per_read, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": None}

    ctx.make_codes_t(pr, sort=True)
    k1_names = {4: "k1_lane4s2_kernel", 5: "k1_lane4_kernel"}
    for kk, dim in ((4, 136), (5, 512)):
        outk = torch.empty((n, dim), dtype=torch.int32, device=dev)
        # warm like the headline kernel: 30 untimed launches, then 100 timed ones (20 launches from an idle chip measured
        # the clock ramp: the line said 0.431 where a warm rocprof mean gives 0.415); the counter child runs keep to a few
        warm = reps >= 10
        for _ in range(30 if warm else 1):
            ctx.kmer_counts4t_dev(pr, out=outk, k=kk)
        t = timed(lambda: ctx.kmer_counts4t_dev(pr, out=outk, k=kk), 10 * reps if warm else 2 * reps)
        assert int(outk[:1024].sum(dim=1).min().item()) == L - kk + 1
        res[f"k1_k{kk}"] = entry(k1_names[kk], t, -(-L // 4) + 4 * dim, n)
        del outk
    pr.codes_t = pr.group_off4 = pr.order4 = None
    m = min(n, 400_000)
    sub = lrb.PackedReads(pr.codes, pr.mask, pr.code_off[: m + 1].contiguous(), pr.mask_off[: m + 1].contiguous(),
                          pr.lens[:m].contiguous(), m)
    half = torch.zeros(lrb.K15_HALF_ENTRIES, dtype=torch.int32, device=dev)
    wl = ctx.lists_alloc(sub, bins=32)
    hist = torch.empty((m, 32), dtype=torch.int32, device=dev)
    sums = torch.empty(m, dtype=torch.int32, device=dev)
    ctx.lists_part_dev(sub, bins=32, out=wl)
    ctx.lists_tally_dev(wl, half)          # (first touch of the workspaces)
    cmap = ctx.cov_map_build_half_dev(half, 10, 32)
    r2 = max(1, reps // 3)
    t_part = timed(lambda: ctx.lists_part_dev(sub, bins=32, out=wl), r2)
    t_tally = timed(lambda: ctx.lists_tally_dev(wl, half), r2)
    t_sweep = timed(lambda: ctx.cov_lists_sweep_dev(wl, cmap, 32, hist=hist, sums=sums), r2)
    assert int(sums.min().item()) == L - 14
    keep_h = hist.clone()
    # K3 as the coverage stage runs it with the library's defaults: the windows counted, parted and ordered AGAIN
    # (lrb_cov_hist_sweep_dev = what lrb_packed_cov_hist_many calls), then the same sweep
    ctx.cov_hist_sweep_dev(sub, cmap, 32, hist=hist, sums=sums)
    t_k3_default = timed(lambda: ctx.cov_hist_sweep_dev(sub, cmap, 32, hist=hist, sums=sums), r2)
    assert torch.equal(keep_h, hist) and int(sums.min().item()) == L - 14
    del keep_h
    res["k2"] = entry(["wl_count_kernel", "wl_gscan_kernel", "wl_part_kernel", "wl_order_kernel", "wl_tally_kernel"], t_part + t_tally,
                      -(-L // 4) + 8 * (L - 14), m)
    res["k2"]["part_and_order_ms"], res["k2"]["tally_ms"] = t_part, t_tally
    k3_bytes = -(-L // 4) + 4 * (L - 14) + 4 * 32
    res["k3_default"] = entry(["wl_count_kernel", "wl_part_kernel", "wl_order_kernel", "wl_map_pack_kernel", "wl_sweep_kernel"],
                              t_k3_default, k3_bytes, m)
    res["k3_default"]["note"] = ("what run_15mer_vecs / the sharded driver's phase B do with the library's defaults "
                                 "(lrb_packed_cov_hist_many -> lrb_cov_hist_sweep_dev): the windows are counted, parted and ordered "
                                 "again, then swept -- K2's lists are gone by then")
    res["k3_kept_lists"] = entry(["wl_map_pack_kernel", "wl_sweep_kernel"], t_sweep, k3_bytes, m)
    res["k3_kept_lists"]["note"] = ("the sweep alone, of window lists K2 left in memory of their own (LRB_KEEP_LISTS=1; their "
                                    "partition passes are K2's), the map packed to 5 bits a pair first: opt-in, see "
                                    "c4_phases.why_default_is_not_kept_lists")
    # the same two routes for a histogram of 64 bins (--bin-count 33..256: the map stays a byte a pair, two buckets of it in
    # LDS, eight entry waves -- the twelve-wave form of that map would spill and is not built)
    del wl, hist, cmap
    if not bins64:   # (the counter runs: one shape per kernel name, so that a mean per launch means one thing)
        del half, sums
        torch.cuda.empty_cache()
        res.update(clustering_stages(torch, lrb, ctx, dev, timed))
        return res
    cmap64 = ctx.cov_map_build_half_dev(half, 10, 64)
    wl64 = ctx.lists_part_dev(sub, bins=64)
    hist64 = torch.empty((m, 64), dtype=torch.int32, device=dev)
    ctx.cov_lists_sweep_dev(wl64, cmap64, 64, hist=hist64, sums=sums)
    t64 = timed(lambda: ctx.cov_lists_sweep_dev(wl64, cmap64, 64, hist=hist64, sums=sums), r2)
    del wl64
    ctx.cov_hist_sweep_dev(sub, cmap64, 64, hist=hist64, sums=sums)
    t64d = timed(lambda: ctx.cov_hist_sweep_dev(sub, cmap64, 64, hist=hist64, sums=sums), r2)
    assert int(sums.min().item()) == L - 14
    k3_bytes64 = -(-L // 4) + 4 * (L - 14) + 4 * 64
    res["k3_bins64"] = {"default": entry(["wl_count_kernel", "wl_part_kernel", "wl_order_kernel", "wl_sweep_kernel"], t64d, k3_bytes64, m),
                        "kept_lists": entry(["wl_sweep_kernel"], t64, k3_bytes64, m),
                        "note": "bin_count 64: byte map (no packing pass), wl_sweep_kernel<4, 8, 3, true, 8, 8>"}
    del half, hist64, sums, cmap64
    torch.cuda.empty_cache()
    res.update(clustering_stages(torch, lrb, ctx, dev, timed))
    if traffic:
        try:
            cc = collect_counters(["--stages-child", "--reads", str(n), "--read-len", str(L), "--no-cpu-baseline",
                                   "--no-extra", "--no-c4", "--no-traffic"])

            def hbm(prefix, every=False):
                """bytes per launch of the first kernel whose name contains `prefix` (every=True: of all of them together:
                a stage made of several kernels, each launched once per call -- wl_order_kernel is two, the register-held
                form _occ1 and the streamed one behind it for the long lists)"""
                f = [v for k_, v in cc["FETCH_SIZE"].items() if prefix in k_]
                w = [v for k_, v in cc["WRITE_SIZE"].items() if prefix in k_]
                if not f or not w:
                    raise RuntimeError(f"no counters for {prefix}")
                if every:
                    return 2.0 * sum(f) * 1024.0 + sum(w) * 1024.0
                return 2.0 * f[0] * 1024.0 + w[0] * 1024.0    # KiB; a 128-byte request is tallied at 64 B on gfx950

            res["k1_k4"]["traffic"] = hbm("k1_lane4s2_kernel")
            res["k1_k5"]["traffic"] = hbm("k1_lane4_kernel")
            for st_ in ("k2", "k3_default", "k3_kept_lists"):
                per = {k_: hbm(k_, every=True) for k_ in res[st_]["kernels"]}
                res[st_]["traffic"], res[st_]["traffic_by_kernel"] = sum(per.values()), per
            for st_, kn in (("k4_seed_hist", "seed_hist_kernel"), ("k5_gauss", "gauss_assign_kernel"),
                            ("k6_core", "hdb_core"), ("k6_mst", "hdb_nearest")):
                try:
                    res[st_]["traffic"] = hbm(kn, every=True)
                except Exception:  # noqa: BLE001 -- a kernel the child run did not see under that name
                    pass
            # (the encode stages share their kernels' names -- the two shapes cannot be told apart in the child run)
            res["traffic_source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child runs of this command (2 x FETCH + WRITE, bytes per launch)"
        except Exception as e:  # noqa: BLE001
            res["traffic_source"] = f"in-run measurement failed ({type(e).__name__}: {e})"
    return res


def clustering_stages(torch, lrb, ctx, dev, timed):
    """The kernels behind the path's last stages, at C1 / C5 sizes on synthetic latents, each with the figure SURVEY
    8(d) prices it by AND the bound it really has:
      k4_seed_hist  cluster_utils.py:136-192: histc(0.5 - M @ M[s], 60, 0, 0.3) for S = 1000 seeds over N = 432,333
                    latents of 4 dims: `4 N latent + 240 S` bytes per fused pass (7 MB: not an HBM story) -- N x S LDS
                    histogram atomics are what it costs
      k5_gauss      cluster_utils.py:261-268,309-322: left-over reads against C cluster Gaussians, fp64 exp / log
      vae_encode    ae_utils.py:141-161: `2 (in 128 + 128 128 + 128 latent)` FLOP per sample against the fp32 MFMA peak
      k6_core/_mst  cluster_utils.py:494 (HDBSCAN): core distances (k = 250) and the spanning tree, 200 k x 8 latents,
                    2 VALU per dimension and pair against the fp32 vector peak
    """
    out = {}
    g = torch.Generator(device=dev).manual_seed(4)
    # ---- K4
    N, d, S = 432_333, 4, 1000
    M = torch.randn((N, d), device=dev, generator=g) * 0.2 + torch.tensor([1.0, 0.3, -0.5, 0.8], device=dev)
    M = (M / M.norm(dim=1, keepdim=True) * (0.5 ** 0.5)).contiguous()      # cluster_utils.normalize: |row|^2 = 0.5
    seeds = torch.randint(0, N, (S,), device=dev, generator=g, dtype=torch.int64)
    hs = ctx.seed_hist_dev(M, seeds)
    ms = timed(lambda: ctx.seed_hist_dev(M, seeds, out=hs), 20)
    by = 4 * N * d + 240 * S
    lanes = float(N) * S
    out["k4_seed_hist"] = {"bound": "lds_atomic", "kernel": "seed_hist_kernel", "kernel_ms": ms, "points": N, "dims": d, "seeds": S,
                           "algorithmic_bytes": by, "achieved_hbm_GBps": by / (ms * 1e-3) / 1e9,
                           "frac_hbm": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "achieved": lanes / (ms * 1e-3) / 1e9, "unit": "G histogram atomics/s (one per point and seed)",
                           # the LDS atomic rate MEASURED on this part: a conflict-free 64-lane ds_add_u32 every 9 clocks
                           # of a CU = 7.1 lanes a clock (wl_count_kernel, one such atomic a window and nothing else that
                           # counts: profiles/r04_k2k3_experiments.txt; scripts/ubench_lds.hip) -- not the 16 lanes a clock
                           # of a 4-clock issue
                           "peak": 256 * (64 / 9.0) * 2.4, "peak_definition": "256 CUs x 64 lanes per 9 clocks (measured rate of a conflict-free ds_add_u32) x 2.4 GHz",
                           "frac": lanes / (ms * 1e-3) / 1e9 / (256 * (64 / 9.0) * 2.4),
                           "valu_bound": {"valu_per_pair": 7.75, "min_kernel_ms": lanes * 7.75 / 64 / 256 / 2.4e9 * 1e3,
                                          "note": "ISA of seed_hist_kernel<4, 4>: 62 VALU a block of eight pairs (packed FP32 for the dot products and "
                                                  "the bin arithmetic), one wave instruction a clock and CU"},
                           "traffic": None}
    del hs
    # ---- K5
    U, F, Cn = 100_000, 42, 8
    X = torch.rand((U, F), device=dev, generator=g, dtype=torch.float64)
    mean = torch.rand((Cn, F), device=dev, generator=g, dtype=torch.float64)
    std = torch.rand((Cn, F), device=dev, generator=g, dtype=torch.float64) * 0.2 + 0.05
    ctx.gauss_assign_dev(X, mean, std)
    ms = timed(lambda: ctx.gauss_assign_dev(X, mean, std), 20)
    by = 8 * F * U + 16 * F * Cn
    fl = float(U) * Cn * F              # one exp / log / divide chain per (read, cluster, feature)
    # a term = log(exp(-z^2 / 2) / (sqrt(2 pi) sigma) + 1e-7) with z = (x - mu) / sigma, all float64 as numpy does it: in the
    # kernel's ISA (hipcc -S, the feature loop of gauss_assign_kernel) 137 fp64 FLOP (46 add, 13 mul, 39 fma / fmac / div_fmas
    # counted twice) in 155 VALU instructions -- two divisions, exp and log expanded by the compiler.  Peak: the fp64 vector rate,
    # 256 CUs x 64 lanes x 2 FLOP x 2.4 GHz = 78.6 TFLOP/s (half the fp32 vector figure of MI355X_MICROARCH.md; one wave64
    # instruction a clock and CU).  A wave takes one read and its lanes the features: F of 64 lanes work.
    flop_per_term, valu_per_term = 137.0, 155.0
    wave_instr = float(U) * Cn * -(-F // 64) * valu_per_term
    out["k5_gauss"] = {"bound": "fp64_valu", "kernel": "gauss_assign_kernel", "kernel_ms": ms, "reads": U, "features": F,
                       "clusters": Cn, "algorithmic_bytes": by, "achieved_hbm_GBps": by / (ms * 1e-3) / 1e9,
                       "frac_hbm": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "terms_per_s": fl / (ms * 1e-3), "flop_per_term": flop_per_term, "valu_per_term": valu_per_term,
                       "achieved": fl * flop_per_term / (ms * 1e-3) / 1e12, "peak": 78.6, "unit": "TFLOP/s fp64 (vector)",
                       "frac": fl * flop_per_term / (ms * 1e-3) / 1e12 / 78.6,
                       "valu_issue_frac": wave_instr / (256 * 2.4e9 * ms * 1e-3),
                       "lanes_active": F / 64.0 if F < 64 else 1.0, "traffic": None}
    del X
    # ---- VAE encode
    from lrbinner_amd import ae_utils
    from lrbinner_amd.vae_native import NativeTrainer
    for name, rows, cov, prof, latent in (("vae_encode", 432_333, 10, 32, 4), ("vae_encode_c3", 1_000_000, 32, 136, 8)):
        data = torch.rand(rows, cov + prof, device=dev)
        vae = ae_utils.VAE(cov, prof, latent_dims=latent, hidden_layers=[128, 128], device="cuda")
        w = ae_utils.h_params[str(prof)]
        tr = NativeTrainer(ctx, vae, 8192, [w["e_cov_weight"], w["e_comp_weight"], w["kld_weight"]])
        try:
            tr.push()
            tr.encode(data)
            ms = timed(lambda: tr.encode(data), 10)
        finally:
            tr.close()
        fl = 2.0 * ((cov + prof) * 128 + 128 * 128 + 128 * latent) * rows
        by = 4.0 * rows * (cov + prof + latent)
        out[name] = {"bound": "mfma", "kernel": "lrb_vae_encode_dev (fused encoder)", "kernel_ms": ms, "rows": rows,
                     "in": cov + prof, "latent": latent, "flop_per_sample": fl / rows,
                     "achieved": fl / (ms * 1e-3) / 1e12, "peak": 157.3, "unit": "TFLOP/s", "frac": fl / (ms * 1e-3) / 1e12 / 157.3,
                     "achieved_hbm_GBps": by / (ms * 1e-3) / 1e9, "frac_hbm": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "note": "fp32 (the reference's precision); rows in + latents out are 4 (in + latent) bytes per sample",
                     "traffic": None}
        del data, vae
    # ---- K6
    F6, d6, k6 = 200_000, 8, 250
    centers = torch.randn((40, d6), device=dev, generator=g) * 2.0
    Xh = (centers[torch.randint(0, 40, (F6,), device=dev, generator=g)] + torch.randn((F6, d6), device=dev, generator=g) * 0.25).contiguous()
    core = ctx.hdb_core_dist_dev(Xh, k6)
    ms_core = timed(lambda: ctx.hdb_core_dist_dev(Xh, k6), 3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, _, _, rounds = ctx.hdb_mst_dev(Xh, core)
    ms_mst = (time.perf_counter() - t0) * 1e3
    pairs = float(F6) * F6
    for name, ms_, sweeps in (("k6_core", ms_core, 5.0), ("k6_mst", ms_mst, float(rounds))):
        fl = pairs * sweeps * 2 * d6           # all-pairs upper bound: the pruned kernels skip most tiles
        out[name] = {"bound": "fp32_valu", "kernel": "hdb_core_sel_kernel" if name == "k6_core" else "hdb_nearest_pruned_kernel (+ host union-find)",
                     "kernel_ms": ms_, "points": F6, "dims": d6, "k": k6, "sweeps": sweeps,
                     "all_pairs_flop": fl, "achieved": fl / (ms_ * 1e-3) / 1e12, "peak": 157.3, "unit": "TFLOP/s (all-pairs equivalent)",
                     "frac": fl / (ms_ * 1e-3) / 1e12 / 157.3,
                     "note": "all-pairs equivalent work / time: above 1.0 means the spatial pruning skipped that share of the pairs",
                     "algorithmic_bytes": 4.0 * d6 * F6 * sweeps, "traffic": None}
    return out


def measure_traffic(kernel_prefix, n, L, k, k1_mode):
    """HBM bytes per launch of the K1 kernel from two rocprofv3 --pmc child runs of this script (one
    counter per run, as MI355X_MICROARCH.md prescribes; the children skip everything but 3 launches)."""
    import csv
    import shutil
    prof = shutil.which("rocprofv3")
    if not prof:
        raise RuntimeError("rocprofv3 not on PATH")
    vals = {}
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [prof, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "k1", "--", sys.executable,
                   os.path.abspath(__file__), "--steps", "3", "--warmup", "1", "--clock-ramp-ms", "0",
                   "--no-cpu-baseline", "--no-extra", "--no-c4", "--no-traffic", "--reads", str(n),
                   "--read-len", str(L), "--k", str(k), "--k1-mode", str(k1_mode)]
            subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=150,
                           env=dict(os.environ, LRB_BENCH_CHILD="1", TMPDIR="/tmp"), cwd="/tmp")
            got = []
            for root, _, files in os.walk(d):
                for fn in files:
                    if fn.endswith("counter_collection.csv"):
                        for r in csv.DictReader(open(os.path.join(root, fn))):
                            if kernel_prefix in r["Kernel_Name"] and r["Counter_Name"] == counter:
                                got.append(float(r["Counter_Value"]))
            if not got:
                raise RuntimeError(f"no {counter} rows for {kernel_prefix}")
            vals[counter] = float(np.mean(got))
    # both counters are in KiB; a wide coalesced streaming read is tallied at half its bytes on gfx950
    return 2.0 * vals["FETCH_SIZE"] * 1024.0 + vals["WRITE_SIZE"] * 1024.0


def c4_phases(torch, dist, lrb, use_dist, dev, rank, world, local, m, L, force_collective=False):
    """BASELINE configs[3] shape (20 M reads over 8 GPUs = 2.5 M per rank): the device-resident core of a rank of the
    sharded driver, timed THROUGH THE PRODUCT'S OWN OBJECTS -- lrbinner_amd.dist.HipCompute, the calls and the order of
    lrbinner_amd.dist.profile_file_sharded:

        resident batches (one per PARSE_CHUNK_BYTES of FASTA, as the parser pool hands them over; made here from
        bases generated on the device: lrb_packed_create_dev)
        K1   Context.kmer_counts_many_dev                         (the kernel half of phase A's kmer_text, all batches at once)
        K2   HipCompute.k15_tally_half_many                       (slice lists -> canonical half of the table)
        ->   dist.allreduce_table (RCCL; 2 GiB)                   (the path's one collective)
        ->   HipCompute.table_from_half                           (expand: the table of the table file)
        K3   HipCompute.cov_hist_groups                           (map build + the kernel half of phase B)

    twice: with the library's defaults (`default`: the coverage phase partitions the windows again) and with the slice
    lists kept across the collective (`kept_lists`: LRB_KEEP_LISTS=1 + the context's list pool, what a long-lived host
    sets; the first pass pays the lists' hipMalloc, reported as first_pass_ms).  `with_text_ms` adds what the driver
    does beyond the kernels on the same objects: K8 formatting + the D2H of the text and its integers -- PCIe-inclusive,
    not part of reads_per_s -- with the composition text (K8 + 4.4 GB of D2H per 2.5 M reads) on a second context's
    stream BESIDE K2, since nothing needs that text before the file is written (`com_text_wait_ms`: what is left of it
    once the table stands); `with_text_serial_*`: the same work batch by batch in front of K2, as phase A of the sharded
    driver interleaves it with parsing (ResidentBatch.kmer_text, HipCompute.cov_text_groups).
    Weak scaling: every rank owns m reads.  Times are max over ranks of the second of two passes."""
    from lrbinner_amd import dist as ld
    collective = world > 1 or force_collective
    ok, err = 1, None
    packed, comp = [], None
    try:
        comp = ld.HipCompute(local)
        per = max(1, ld.PARSE_CHUNK_BYTES // (L + 8))      # reads of one parser range (~6,700 at 10 kb)
        letters = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
        g = torch.Generator(device=dev).manual_seed(777 + rank)
        for a in range(0, m, per):
            nb = min(per, m - a)
            seqs = letters[torch.randint(0, 4, (nb * L,), device=dev, generator=g, dtype=torch.int64)]
            rb = comp.ctx.packed_create_dev(seqs.data_ptr(), np.arange(nb + 1, dtype=np.uint64) * np.uint64(L), with_planes=2)
            packed.append(ld._HipPacked(rb))
            del seqs
        items = list(enumerate(packed))
        counts = torch.empty((m, 136), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001
        ok, err = 0, f"{type(e).__name__}: {e}"

    def agree(ok_, err_):
        # a rank that dropped out alone would leave the others waiting in the next collective: fail together
        if use_dist:
            flag = torch.tensor([ok_], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0 and ok_:
                return 0, "another rank failed"
        return ok_, err_

    ok, err = agree(ok, err)
    if not ok:
        for p_ in packed:
            p_.free()
        return {"error": err}

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # the collective as the sharded driver issues it (dist.allreduce_table: torch.distributed's all_reduce, or
    # lrb_k15_allreduce on a communicator made through the C ABI with LRB_COLLECTIVE=abi); with one rank that function
    # returns at once, so the forced single-rank form calls torch.distributed directly
    allreduce = (lambda t: dist.all_reduce(t)) if world == 1 else (lambda t: ld.allreduce_table(t, None, comp))

    def one_pass(text=False):
        ph, state = {}, {}
        half = comp.new_half()
        fence()
        t_start = time.perf_counter()

        def lap(name, fn):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ph[name] = (time.perf_counter() - t0) * 1e3

        def k1():
            if text != "serial":   # the tallies of all resident batches behind one launch (lrb_packed_kmer_counts_many_dev)
                comp.ctx.kmer_counts_many_dev([p_.rb for p_ in packed], 4, counts.data_ptr())
                return
            for p_ in packed:
                p_.kmer_text(4)          # K1 + K8 + D2H of text and integers, batch by batch, as phase A calls it

        # text == True: the composition text is not needed before the file is written, so K8 + the D2H of text and integers
        # (4.4 GB per 2.5 M reads: PCIe) run on a SECOND context's stream, from a thread of their own, while this thread
        # goes on to K2 -- the tallies are in HBM, the formatter only reads them
        side = {"thread": None, "err": None}

        def text_behind():
            try:
                from lrbinner_amd._lib import call, vp
                for a in range(0, m, TEXT_ROWS):
                    b = min(m, a + TEXT_ROWS)
                    call("lrb_format_com_dev", ctx2._h, vp(counts[a:b].data_ptr()), vp(lens_all[a:b].data_ptr()), b - a, 136, 4,
                         vp(d_text.data_ptr()), vp(d_q.data_ptr()))
                    ctx2.d2h(p_text[: (b - a) * 1225], d_text.data_ptr())
                    ctx2.d2h(p_q[: (b - a) * 136], d_q.data_ptr())
            except BaseException as e:  # noqa: BLE001 -- raised on the main thread after the join
                side["err"] = e

        def k3():
            if text:
                for _ in comp.cov_text_groups(items, state["table"], 10, 32, kept=state["kept"]):
                    pass
            else:
                for _ in comp.cov_hist_groups(items, state["table"], 10, 32, kept=state["kept"]):
                    pass

        lap("k1_k4_ms", k1)
        if text is True:
            import threading
            side["thread"] = threading.Thread(target=text_behind)
            side["thread"].start()
        lap("k2_ms", lambda: state.update(kept=comp.k15_tally_half_many(packed, half, keep_bins=32)))
        if collective:
            lap("allreduce_ms", lambda: allreduce(half))
        lap("expand_ms", lambda: state.update(table=comp.table_from_half(half)))
        state["kept_groups"] = len(state["kept"])
        if side["thread"] is not None:   # what is left of the composition text once the table stands
            lap("com_text_wait_ms", side["thread"].join)
            if side["err"] is not None:
                raise side["err"]
        lap("k3_ms", k3)
        fence()
        ph["total_ms"] = (time.perf_counter() - t_start) * 1e3
        # every slot of the table counts both strands of every rank's reads
        total = int(state["table"].to(torch.int64).bitwise_and(0xFFFFFFFF).sum().item())
        good = total == 2 * world * m * (L - 14)
        why = None if good else f"table total {total} != {2 * world * m * (L - 14)}"
        if text != "serial":
            rows = counts.sum(dim=1)
            k1_ok = int(rows.min().item()) == L - 3 == int(rows.max().item())
            if not k1_ok:
                bad = torch.nonzero(rows != L - 3).flatten()
                why = (why or "") + f" K1 row sums: {bad.numel()} of {m} rows off, first {bad[:4].tolist()} = {rows[bad[:4]].tolist()}"
            good = good and k1_ok
        del state["table"], half
        return ph, (good, why), state["kept_groups"]

    TEXT_ROWS = 65536
    ctx2 = lrb.Context(local)                       # a context (and a stream) of its own for the text leg
    lens_all = torch.full((m,), L, dtype=torch.int32, device=dev)
    d_text = torch.empty(TEXT_ROWS * 1225 + 16, dtype=torch.uint8, device=dev)
    d_q = torch.empty((TEXT_ROWS, 136), dtype=torch.int32, device=dev)
    p_text = ctx2.pinned("c4_text", TEXT_ROWS * 1225)
    p_q = ctx2.pinned("c4_q", TEXT_ROWS * 136 * 4, np.uint32)
    torch.cuda.synchronize()

    def route(keep):
        os.environ["LRB_KEEP_LISTS"] = "1" if keep else "0"
        comp.ctx.list_pool((160 << 30) if keep else 0)
        try:
            first, (good0, why0), _ = one_pass()
            ph, (good, why1), kept_groups = one_pass()
            ph_text, (good2, why2), _ = one_pass(text=True)
            ph_serial, (good3, why3), _ = one_pass(text="serial")
            good = good0 and good and good2 and good3
            err_ = None if good else f"result check failed on rank {rank}: first pass {why0}; second {why1}; text pass {why2}; serial text pass {why3}"
        except Exception as e:  # noqa: BLE001
            good, err_ = False, f"{type(e).__name__}: {e}"
        good, err_ = agree(1 if good else 0, err_)
        if not good:
            return {"error": err_}
        keys = sorted(ph)
        tkeys = sorted(ph_text)
        v = torch.tensor([ph[k_] for k_ in keys] + [ph_text[k_] for k_ in tkeys] + [ph_serial["total_ms"], first["total_ms"]], dtype=torch.float64, device=dev)
        per_rank = None
        if use_dist:
            every = [torch.empty_like(v) for _ in range(world)]
            dist.all_gather(every, v)          # every rank's own phase times (the line reports them beside the maxima)
            per_rank = {k_: [round(float(e[i].item()), 3) for e in every] for i, k_ in enumerate(keys)}
            dist.all_reduce(v, op=dist.ReduceOp.MAX)
        v = v.tolist()
        ph = {k_: float(x) for k_, x in zip(keys, v[:len(keys)])}
        ph_text = {k_: float(x) for k_, x in zip(tkeys, v[len(keys):len(keys) + len(tkeys)])}
        return {"phases_ms_max_over_ranks": ph, "phases_ms_per_rank": per_rank, "reads_per_s": m * world / (ph["total_ms"] * 1e-3),
                "groups_with_kept_lists": kept_groups, "first_pass_ms": float(v[-1]),
                "with_text_ms": ph_text, "with_text_reads_per_s": m * world / (ph_text["total_ms"] * 1e-3),
                "with_text_serial_total_ms": float(v[-2]), "with_text_serial_reads_per_s": m * world / (float(v[-2]) * 1e-3)}

    saved = os.environ.get("LRB_KEEP_LISTS")
    try:
        routes = {"default": route(False), "kept_lists": route(True)}
    finally:
        comp.ctx.list_pool(0)
        if saved is None:
            os.environ.pop("LRB_KEEP_LISTS", None)
        else:
            os.environ["LRB_KEEP_LISTS"] = saved
        for p_ in packed:
            p_.free()
        comp.ctx.trim()
        ctx2.close()
    if "error" in routes["default"]:
        return {"error": routes["default"]["error"], "routes": routes}
    res = {"workload": f"{m} synthetic {L}-base reads per GPU in {len(packed)} resident batches, k=4 + 15-mer table + coverage "
                       f"(bin_size 10, 32 bins); BASELINE configs[3] shape ({m * world} reads over {world} GPU(s))",
           "timed_through": "lrbinner_amd.dist.HipCompute (k15_tally_half_many, table_from_half, cov_hist_groups), "
                            "dist.allreduce_table, ResidentBatch.kmer_counts_dev -- the objects and call order of profile_file_sharded",
           "reads_per_gpu": m, "world_size_seen_by_rccl": dist.get_world_size() if use_dist else 1,
           "allreduce": "half" if collective else "none",
           "collective_via": os.environ.get("LRB_COLLECTIVE", "torch") if collective else None,
           "routes": routes,
           # the headline of this block is what the product does with its defaults
           "phases_ms_max_over_ranks": routes["default"]["phases_ms_max_over_ranks"],
           "reads_per_s": routes["default"]["reads_per_s"], "reads_per_s_route": "default",
           "why_default_is_not_kept_lists": "keeping a group's lists needs 17.6 GB of their own: hipMalloc of that size takes 0.13-0.48 s "
                                            "once ~50 GB are allocated (profiles/r05_side_alloc.txt, r05_c3_stage_calls.txt) against the "
                                            "13.5 ms partition pass it saves, and on a side thread it holds up the main thread's "
                                            "allocations (composition stage 0.52 -> 2.9 s); a host that runs call after call sets "
                                            "LRB_KEEP_LISTS=1 + lrb_ctx_list_pool and pays it once (`kept_lists.first_pass_ms`)",
           "scaling": "weak"}
    ph = res["phases_ms_max_over_ranks"]
    if "allreduce_ms" in ph:
        nbytes = lrb.K15_HALF_ENTRIES * 4
        res["allreduce_bytes"] = nbytes
        res["allreduce_algbw_GBps"] = nbytes / (ph["allreduce_ms"] * 1e-3) / 1e9
        res["allreduce_busbw_GBps"] = res["allreduce_algbw_GBps"] * 2 * (world - 1) / world  # 0 with one rank
        res["allreduce_ms"] = ph["allreduce_ms"]
        # SURVEY 8(e)'s cost model beside it: xGMI is point to point (7 links a GPU, ~75 GB/s a direction usable of
        # ~153 GB/s); a direct reduce-scatter + all-gather moves (P-1)/P of the buffer per phase over P-1 links at once,
        # a ring moves 2 (P-1)/P of it over ONE link
        p = max(world, 1)
        link = 75e9
        res["allreduce_cost_model"] = {
            "direct_ms": 0.0 if p == 1 else 2 * (nbytes / p) / link * 1e3,
            "ring_ms": 0.0 if p == 1 else 2 * (p - 1) / p * nbytes / link * 1e3,
            "link_GBps_per_direction": link / 1e9, "links_per_gpu": 7,
            "note": "direct = every peer pair on its own xGMI link (reduce-scatter + all-gather of 1/P slices); ring = one link at a time"}
    return res


def extra_stages(torch, dist, lrb, ctx, pr, use_dist, dev, m, L):
    """Secondary numbers: K2 accumulate, mirror, (all-reduce), K3 on the first m reads."""
    sub = lrb.PackedReads(pr.codes, pr.mask, pr.code_off[: m + 1].contiguous(),
                          pr.mask_off[: m + 1].contiguous(), pr.lens[:m].contiguous(), m)
    table = torch.zeros(lrb.K15_ENTRIES, dtype=torch.int32, device=dev)

    def timed(fn):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b)

    res = {"sample_reads": m}
    t = timed(lambda: ctx.k15_accumulate_dev(sub, table))
    res["k2_direct_atomics_ms"] = t
    # (the forward table of these reads by single atomics: what the mirror and the gather-form K3 below start from; the
    # product's K2 -- window lists into the canonical half -- is roofline_stages.k2)
    res["k2_mirror_ms"] = timed(lambda: ctx.k15_mirror_dev(table))
    hist = torch.empty((m, 32), dtype=torch.int32, device=dev)
    sums = torch.empty(m, dtype=torch.int32, device=dev)
    t = timed(lambda: ctx.cov_hist_dev(sub, table, 10, 32, hist=hist, sums=sums))
    res["k3_ms"] = t
    res["k3_reads_per_s"] = m / (t * 1e-3)
    assert int(sums.min().item()) == L - 14
    # the same histograms from the compact map of the table (bin ids of the canonical half, 512 MB)
    keep = hist.clone()
    cmap = ctx.cov_map_build_dev(table, 10, 32)
    ctx.cov_hist_map_dev(sub, cmap, 32, hist=hist, sums=sums)
    t = timed(lambda: ctx.cov_hist_map_dev(sub, cmap, 32, hist=hist, sums=sums))
    assert torch.equal(keep, hist)
    res["k3_map_ms"] = t
    res["k3_map_reads_per_s"] = m / (t * 1e-3)
    # every gather is one 128-byte line fill (rocprofv3 TCC_EA0_RDREQ_128B = 1.00 per gather,
    # profiles/r02_k3_rocprof_summary.txt): the memory system moves this much for K3
    res["k3_map_line_fill_GBps"] = m * (L - 14) * 128 / (t * 1e-3) / 1e9
    # ... and as a sweep: the windows go to the map (partitioned by 2 MB slice, 4 + 4 streamed bytes per window)
    # instead of a 128-byte line per window coming to them; on all the resident reads in 400 k-read groups
    n_all = pr.n
    parts, step = [], 400_000
    for a in range(0, n_all, step):
        b = min(n_all, a + step)
        parts.append(lrb.PackedReads(pr.codes, pr.mask, pr.code_off[a:b + 1].contiguous(), pr.mask_off[a:b + 1].contiguous(),
                                     pr.lens[a:b].contiguous(), b - a))
    hist_all = torch.empty((n_all, 32), dtype=torch.int32, device=dev)
    sums_all = torch.empty(n_all, dtype=torch.int32, device=dev)

    def sweep_all():
        a = 0
        for p_ in parts:
            ctx.cov_hist_sweep_dev(p_, cmap, 32, hist=hist_all[a:a + p_.n], sums=sums_all[a:a + p_.n])
            a += p_.n

    sweep_all()
    t = timed(sweep_all)
    assert torch.equal(keep, hist_all[:m]) and int(sums_all.min().item()) == L - 14
    res["k3_sweep_ms"] = t
    res["k3_sweep_reads"] = n_all
    res["k3_sweep_reads_per_s"] = n_all / (t * 1e-3)
    res["k3_sweep_roofline_frac"] = (-(-L // 4) + 4 * (L - 14) + 4 * 32) * n_all / (t * 1e-3) / 1e9 / HBM_PEAK_GBS  # 42,572 B per read
    del cmap, keep, hist_all, sums_all
    del table
    return res


if __name__ == "__main__":
    main()
