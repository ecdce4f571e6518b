"""Multi-GPU form of the profile path: one process per GPU (torch.distributed, backend
"nccl" = RCCL over xGMI), reads sharded across ranks, ONE collective.

The path shards naturally: composition (K1) and coverage (K3) are per
read; the 15-mer table is a sum over reads.  So every rank

    phase A   K1 on its reads; K2 accumulate of its reads into its own 4 GiB table
    exchange  the table T = sum over ranks of (F_r + F_r o rc) is determined by its CANONICAL HALF
              (of x and rc(x) the one whose middle base has high code bit 0): every rank folds its
              forward tallies into that half (2 GiB), ONE all-reduce(sum) of the half -- uint32
              wrap-around is associative and commutative, so the result is bit-identical to the
              serial table -- and expands it again (``reduce_and_mirror``).  The full-table form
              (all-reduce 4 GiB, then mirror) stays behind LRB_ALLREDUCE=full for A/B runs.
    phase B   K3 on its reads against the full table

and rows are written back in input order.  There is no other communication.

Two sharding schemes, both order-preserving:
  * ``shard_range``  -- contiguous index ranges, for data already resident (bench,
    device-level API): concatenating the ranks' outputs reproduces the input order.
  * range-cyclic     -- ``profile_file_sharded`` cuts a plain FASTA file into byte ranges and
    rank r parses ranges r, r+P, r+2P, ... (``lrb_preader_open_shard``: nobody reads what it
    does not own; gzip / FASTQ input cannot be cut, there every rank streams the file and
    keeps every P-th batch).  No rank needs to know the read count up front: profile rows have a
    fixed width, so one all_gather of the batches' read counts after the parse gives every batch
    its first row, and every rank writes its own rows at their final place in ONE file per
    profile (``_ShardWriter``; no part files, nothing for rank 0 to stitch).  A rank's batches
    stay packed in HBM between the composition/accumulate phase and the coverage phase.

The compute object is the GPU context in production (``HipCompute``); tests pass a
small CPU stand-in so the sharding and the collective are exercised under gloo with
world_size 2 without a GPU.
"""
import os

import numpy as np


def shard_range(n, rank, world):
    """Contiguous [lo, hi) of rank ``rank``; the ranges of all ranks tile [0, n)."""
    return (n * rank) // world, (n * (rank + 1)) // world


def _dist():
    import torch.distributed as dist
    return dist


def world_info(group=None):
    dist = _dist()
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


_abi_comms = {}  # process group (the object itself; None = the default group) -> ncclComm_t made through the C ABI


def destroy_abi_comms():
    """Give the communicators made through the C ABI back (lrb_rccl_comm_destroy); called before the process group
    goes and at interpreter exit."""
    from . import device as lrb
    while _abi_comms:
        _, comm = _abi_comms.popitem()
        try:
            lrb.Context.rccl_comm_destroy(comm)
        except Exception:  # noqa: BLE001 -- the runtime may already be shutting down
            pass


import atexit  # noqa: E402
atexit.register(destroy_abi_comms)


def allreduce_table(table_t, group=None, compute=None):
    """In-place sum of the 15-mer table (or its canonical half) over all ranks.  The tensor is viewed
    as int32: two's-complement addition is the same bit pattern as uint32 wrap-around.
    LRB_COLLECTIVE=abi sends it through the library's own entry point (lrb_k15_allreduce on a
    communicator made with lrb_rccl_comm_create, the 128-byte id broadcast through the process
    group) -- the call a non-Python host makes; the default is torch.distributed's all_reduce on the
    same RCCL."""
    import torch
    rank, world = world_info(group)
    if world == 1:
        return table_t
    dist = _dist()
    if (os.environ.get("LRB_COLLECTIVE", "torch") == "abi" and compute is not None and hasattr(compute, "ctx")
            and table_t.is_cuda):
        key = group
        if key not in _abi_comms:
            box = [compute.ctx.rccl_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0, group=group)
            _abi_comms[key] = compute.ctx.rccl_comm_create(world, rank, box[0])
        compute.ctx.k15_allreduce(_abi_comms[key], table_t)
        return table_t
    dist.all_reduce(table_t.view(torch.int32), op=dist.ReduceOp.SUM, group=group)
    return table_t


def allreduce_mode():
    """'half' (default): all-reduce the canonical half of the table; 'full': the whole table."""
    return "full" if os.environ.get("LRB_ALLREDUCE", "half").lower() == "full" else "half"


def reduce_and_mirror(table_t, compute, group=None, timings=None):
    """Forward tallies of every rank -> the full table T[x] = sum_r (F_r[x] + F_r[rc(x)]) on every
    rank, in place.  One rank: just the mirror pass.  ``timings`` (dict) receives the seconds of
    'fold', 'allreduce', 'expand' / 'mirror' when given (the caller synchronises the device)."""
    import time
    rank, world = world_info(group)

    def lap(name, fn):
        if timings is None:
            return fn()
        compute.sync()
        t0 = time.perf_counter()
        r = fn()
        compute.sync()
        timings[name] = timings.get(name, 0.0) + time.perf_counter() - t0
        return r

    if world > 1 and allreduce_mode() == "half" and hasattr(compute, "k15_fold_half"):
        half = lap("fold", lambda: compute.k15_fold_half(table_t))
        lap("allreduce", lambda: allreduce_table(half, group, compute))
        lap("expand", lambda: compute.k15_expand_half(half, table_t))
        if timings is not None:
            timings["allreduce_bytes"] = half.numel() * 4
        return table_t
    if world > 1:
        lap("allreduce", lambda: allreduce_table(table_t, group, compute))
        if timings is not None:
            timings["allreduce_bytes"] = table_t.numel() * 4
    lap("mirror", lambda: compute.k15_mirror(table_t))
    return table_t


class _HipPacked:
    """A batch left packed in HBM (lrb_packed): the three stages run on it without another
    parse or PCIe crossing."""

    def __init__(self, rb):
        self.rb, self.n, self.device_bytes = rb, rb.n, rb.device_bytes
        self.lens = rb.lens

    def kmer_counts(self, k):
        return self.rb.kmer_counts(k)

    def k15_accumulate(self, table):
        self.rb.k15_accumulate(table.data_ptr())

    def cov_hist(self, table, bin_size, bins):
        return self.rb.cov_hist(table.data_ptr(), bin_size, bins)

    # the same stages ending in the text rows, formatted on the device (K8): (text, six-decimal integers)
    def kmer_text(self, k, slot=0):
        return self.rb.kmer_text(k, want_q=True, slot=slot)

    def cov_text(self, table, bin_size, bins, slot=0):
        return self.rb.cov_text(table.data_ptr(), bin_size, bins, want_q=True, slot=slot)

    def free(self):
        self.rb.free()


class HipCompute:
    """The GPU side of one rank: a device context plus a table living in a torch tensor
    (torch owns the allocation so that torch.distributed can reduce it)."""

    def __init__(self, device_index):
        import torch
        from . import device as lrb
        self.torch, self.lrb = torch, lrb
        self.dev = torch.device("cuda", device_index)
        torch.cuda.set_device(self.dev)
        self.ctx = lrb.Context(device_index, use_torch_stream=True)

    def new_table(self):
        return self.torch.zeros(self.lrb.K15_ENTRIES, dtype=self.torch.int32, device=self.dev)

    def resident_budget(self):
        """Bytes of packed reads this rank may keep in HBM between the two phases."""
        free, _ = self.torch.cuda.mem_get_info(self.dev)
        return int(free * 0.6)

    host_packs = True   # the parser pool packs the ranges on the host: pack() takes the HostPacked batch

    def pack(self, seqs, offs, k, hp=None):
        if hp is not None:   # 0.375 bytes a base over PCIe instead of 1
            return _HipPacked(self.ctx.packed_create_packed(hp, with_planes=(1 if k == 3 else 2)))
        return _HipPacked(self.ctx.packed_create(seqs, offs, with_planes=(1 if k == 3 else 2)))

    def kmer_counts(self, seqs, offs, k):
        return self.ctx.kmer_counts(seqs, offs, k)

    def k15_accumulate(self, seqs, offs, table):
        self.ctx.k15_accumulate(seqs, offs, table.data_ptr())

    def k15_accumulate_many(self, packed, table):
        self.ctx.k15_accumulate_many([p.rb for p in packed], table.data_ptr())

    # the table as its canonical half (one counter per pair x / rc(x)): what the ranks all-reduce, born folded
    def new_half(self):
        return self.torch.zeros(self.lrb.K15_HALF_ENTRIES, dtype=self.torch.int32, device=self.dev)

    def k15_tally_half_many(self, packed, half, keep_bins=None):
        """K2 of resident batches from slice lists (groups of batches share a partition); returns the lists kept
        for the coverage phase, {ids of a group's batches: PackedLists}, while a third of the free HBM covers them."""
        from . import runners_utils as ru
        kept = {}
        rbs = [p.rb for p in packed]
        if not (keep_bins and os.environ.get("LRB_KEEP_LISTS", "0") == "1"):
            # (lists of their own memory only on request: allocating 17.6 GB costs thirty times the partition pass it
            # saves a one-shot run -- runners_utils.run_15mer_counts, profiles/r05_side_alloc.txt)
            # (the LAST group's lists stay in the context's workspaces: cov_hist_groups sweeps that group first)
            self.ctx.k15_tally_half_many(rbs, half.data_ptr(), bins=keep_bins or 32)
            return kept
        for group, bases in ru._batch_groups(rbs, ru.SWEEP_GROUP_BASES):
            own = bool(bases >= max(ru.SWEEP_MIN_BASES, ru.K2_LISTS_MIN_BASES) and
                       bases * 14 < self.torch.cuda.mem_get_info(self.dev)[0])
            if not own:
                self.ctx.k15_tally_half_many(group, half.data_ptr())
                continue
            wl = self.lrb.PackedLists(self.ctx, group, min(int(keep_bins), 145), workspace=False)
            wl.tally(half.data_ptr())
            if wl.fits(keep_bins):
                kept[tuple(id(rb) for rb in group)] = wl
            else:
                self.torch.cuda.synchronize()
                wl.free()
        return kept

    def k15_tally_half_one(self, seqs, offs, half):
        """K2 of a batch that does not stay resident."""
        rb = self.ctx.packed_create(seqs, offs, with_planes=0)
        try:
            self.k15_tally_half_many([_HipPacked(rb)], half)
            self.torch.cuda.synchronize()
        finally:
            rb.free()

    def table_from_half(self, half):
        table = self.torch.empty(self.lrb.K15_ENTRIES, dtype=self.torch.int32, device=self.dev)
        self.ctx.k15_expand_half_dev(half, table)
        return table

    def cov_hist_groups(self, items, table, bin_size, bins, kept=None):
        """The KERNEL half of the coverage phase, group by group: K3 of resident batches as a sweep over the compact map
        of the table, several batches per call -- the lists the table phase left (PackedLists.cov_hist: the sweep alone)
        or partition + sweep (lrb_packed_cov_hist_many), as run_15mer_vecs does it.  Yields (group, rows) once a
        group's kernels are enqueued: group = its [(batch id, packed)], rows(slot) = the generator of the text half,
        (batch id, cov_profs text, six-decimal integers, staging slot) per batch -- formatted on the device from the
        histograms the kernels left in the context, so it has to be drained (or dropped) before the next group.
        The map (a pass over the 4 GiB table and 512 MB) is built when the first group takes the sweep; groups below
        SWEEP_MIN_BASES go through the per-batch gather kernel inside rows() and never ask for it.
        bench.py's c4_phases walks this generator without calling rows(): the kernels alone."""
        from . import runners_utils as ru
        cmap = None
        try:
            group, bases = [], 0

            def flush():
                nonlocal cmap
                wl = (kept or {}).pop(tuple(id(p.rb) for _, p in group), None)
                if group and (wl is not None or bases >= ru.SWEEP_MIN_BASES):
                    if cmap is None:
                        cmap = self.ctx.cov_map_build(table.data_ptr(), bin_size, bins)
                    rbs = [p.rb for _, p in group]
                    if wl is not None and wl.fits(bins):
                        wl.cov_hist(cmap, bins)
                    else:
                        self.ctx.cov_hist_many(rbs, cmap, bins)
                    mine = list(group)

                    def rows(slot=0):
                        for (b, _), (s_, txt, q) in zip(mine, self.ctx.cov_rows(rbs, int(bins), True, slot)):
                            yield b, txt, q, s_

                    yield mine, rows
                    if wl is not None:
                        self.torch.cuda.synchronize()
                        wl.free()
                elif group:
                    mine = list(group)

                    def rows(slot=0):
                        for b, p in mine:
                            s_ = slot() if callable(slot) else slot
                            yield (b,) + tuple(p.cov_text(table, bin_size, bins, slot=s_)) + (s_,)

                    yield mine, rows

            # groups as k15_tally_half_many formed them (filled from the end), so that its lists are found again; the
            # last group first when its lists are still in the workspaces (every group after it partitions over them)
            groups = [g for g, _ in ru._batch_groups(list(items), ru.SWEEP_GROUP_BASES, bases_of=lambda it: it[1].rb.total_bases)]
            if len(groups) > 1 and not kept and self.ctx.lists_resident([p.rb for _, p in groups[-1]], bins):
                groups = [groups[-1]] + groups[:-1]
            for g in groups:
                group, bases = list(g), sum(p.rb.total_bases for _, p in g)
                yield from flush()
        finally:
            if cmap is not None:
                self.ctx.free(cmap)
            for wl in (kept or {}).values():
                wl.free()

    def cov_text_groups(self, items, table, bin_size, bins, kept=None, slot=0):
        """(batch id, cov_profs text, six-decimal integers, staging slot) of resident batches: cov_hist_groups' kernels
        and, group by group, their text half.  ``slot``: the staging slot to format into, or a callable handing out the
        slot for the next batch."""
        for _, rows in self.cov_hist_groups(items, table, bin_size, bins, kept=kept):
            yield from rows(slot)

    def k15_mirror(self, table):
        self.ctx.k15_mirror_dev(table)
        self.torch.cuda.synchronize()

    def k15_fold_half(self, table):
        return self.ctx.k15_fold_half_dev(table)

    def k15_expand_half(self, half, table):
        self.ctx.k15_expand_half_dev(half, table)

    def sync(self):
        self.torch.cuda.synchronize()

    def cov_hist(self, seqs, offs, table, bin_size, bins):
        return self.ctx.cov_hist(seqs, offs, table.data_ptr(), bin_size, bins)


def profile_reads_sharded(seqs, offs, k, bin_size, bins, compute, group=None):
    """Array-level sharded profile of reads every rank can see (contiguous shards).
    Returns this rank's (lo, hi, counts[hi-lo, dim], hist[hi-lo, bins], sums[hi-lo])."""
    rank, world = world_info(group)
    n = len(offs) - 1
    lo, hi = shard_range(n, rank, world)
    sub_offs = np.ascontiguousarray(offs[lo:hi + 1])
    counts = compute.kmer_counts(seqs, sub_offs, k)
    table = compute.new_table()
    compute.k15_accumulate(seqs, sub_offs, table)
    reduce_and_mirror(table, compute, group)
    hist, sums = compute.cov_hist(seqs, sub_offs, table, bin_size, bins)
    return lo, hi, counts, hist, sums


def gather_rows(local, group=None):
    """Concatenate per-rank row blocks in rank order on every rank (small results)."""
    rank, world = world_info(group)
    if world == 1:
        return local
    parts = [None] * world
    _dist().all_gather_object(parts, local, group=group)
    return np.concatenate(parts, axis=0)


def _q6_of_values(vals):
    """Six-decimal integers of the float64 values the host formatters return (k / 1e6, correctly rounded)."""
    return np.rint(np.asarray(vals, dtype=np.float64) * 1e6).astype(np.uint32)


def _pwrite_all(fd, buf, at):
    """os.pwrite until every byte of buf (any buffer object) is in the file at offset `at`."""
    mv = memoryview(buf).cast("B")
    done = 0
    while done < len(mv):
        done += os.pwrite(fd, mv[done:], at + done)


class _ShardWriter:
    """Every rank writes ITS rows of the profiles at their final place -- no part files, nothing for rank 0 to stitch.

    Profile rows have a fixed width (every value is "%f" of a ratio in [0, 1]: lrb_com_row_bytes / lrb_cov_row_bytes), so
    a batch's rows belong at first_row(batch) x row_bytes of the text file and first_row x 4 cols of the side-car of
    six-decimal integers ({path}.q6, runners_utils._ValueSidecar) -- once the read counts of the batches in front of it
    are known, which is after every rank has parsed its share (``set_layout``, from one all_gather of the counts).
    Rows that arrive before that wait in memory (``buffer_bytes`` at most, then in a spill file of this rank that
    this rank copies into place itself).  The writes happen on a thread of this object (os.pwrite releases the GIL):
    ``slot()`` hands out one of two staging slots whose previous contents have been consumed, ``put`` queues what was
    formatted into it -- the protocol of runners_utils._ProfileWriter."""

    def __init__(self, rank, buffer_bytes=None):
        import queue
        import threading
        self.rank = rank
        # rows waiting for the layout: 16 GB of host memory for the JOB by default, shared out over the ranks of the node
        world = max(1, int(os.environ.get("WORLD_SIZE", "1")))
        self.cap = int(float(os.environ.get("LRB_DIST_BUFFER_GB", str(16.0 / world))) * (1 << 30)) if buffer_bytes is None else int(buffer_bytes)
        self.prof = {}
        self.first_row = None
        self.held = 0            # bytes waiting in memory for the layout
        self.q = queue.Queue()
        self.free = [threading.Semaphore(1), threading.Semaphore(1)]
        self.turn = 0
        self.err = None
        self.stats = {"buffered_bytes": 0, "spilled_bytes": 0, "direct_bytes": 0}
        self.th = threading.Thread(target=self._run, daemon=True)
        self.th.start()

    def add(self, name, path, row_bytes, cols):
        self.prof[name] = {"path": path, "row_bytes": int(row_bytes), "cols": int(cols), "fd": None, "fdq": None,
                           "held": [], "spill": None, "spilled": []}

    def slot(self):
        s = self.turn
        self.turn ^= 1
        self.free[s].acquire()
        return s

    def put(self, name, b, n_rows, text, q, slot=None):
        """Rows of batch b (file order) of profile `name`: text (bytes / uint8 array of n_rows x row_bytes bytes) and
        the uint32 six-decimal integers [n_rows, cols].  With ``slot`` the arrays are the staging of that slot and are
        released for reuse once written or copied; without, they must stay valid until close()."""
        self.q.put(("rows", name, int(b), int(n_rows), text, q, slot))

    def set_layout(self, first_row):
        """first_row: {batch: index of its first row in the whole file}; the files exist at their full size."""
        self.q.put(("layout", dict(first_row)))

    def close(self):
        self.q.put(None)
        self.th.join()
        for p in self.prof.values():
            for key in ("fd", "fdq"):
                if p[key] is not None:
                    os.close(p[key])
                    p[key] = None
        if self.err is not None:
            raise self.err

    # ---- the writer thread ----
    def _place(self, p, b, n_rows, text, q):
        row = self.first_row[b]
        _pwrite_all(p["fd"], text, row * p["row_bytes"])
        _pwrite_all(p["fdq"], q, row * p["cols"] * 4)

    def _rows(self, name, b, n_rows, text, q):
        p = self.prof[name]
        tb = n_rows * p["row_bytes"]
        if memoryview(text).nbytes != tb or np.asarray(q).size != n_rows * p["cols"]:
            raise ValueError(f"{p['path']}: batch {b} is not {n_rows} rows of {p['row_bytes']} bytes "
                             f"({memoryview(text).nbytes} bytes of text, {np.asarray(q).size} integers)")
        q = np.ascontiguousarray(q, dtype=np.uint32)
        if n_rows == 0:
            return
        if self.first_row is not None:
            self._place(p, b, n_rows, text, q)
            self.stats["direct_bytes"] += tb + q.nbytes
        elif self.held + tb + q.nbytes <= self.cap:
            p["held"].append((b, n_rows, bytes(text), q.tobytes()))
            self.held += tb + q.nbytes
            self.stats["buffered_bytes"] += tb + q.nbytes
        else:
            if p["spill"] is None:
                p["spill"] = open(f"{p['path']}.rank{self.rank}.spill", "w+b")
            f = p["spill"]
            at = f.tell()
            f.write(text)
            f.write(q.tobytes())
            p["spilled"].append((b, n_rows, at))
            self.stats["spilled_bytes"] += tb + q.nbytes

    def _layout(self, first_row):
        self.first_row = first_row
        for p in self.prof.values():
            # (under their .partial names until every rank's rows are in: _finish_profile_files)
            p["fd"] = os.open(p["path"] + ".partial", os.O_WRONLY)
            p["fdq"] = os.open(p["path"] + ".q6.partial", os.O_WRONLY)
            for b, n_rows, text, qb in p["held"]:
                self._place(p, b, n_rows, text, qb)
            p["held"] = []
            if p["spill"] is not None:
                f = p["spill"]
                f.flush()
                for b, n_rows, at in p["spilled"]:
                    tb, qb = n_rows * p["row_bytes"], n_rows * p["cols"] * 4
                    f.seek(at)
                    self._place(p, b, n_rows, f.read(tb), f.read(qb))
                f.close()
                os.remove(f.name)
                p["spill"], p["spilled"] = None, []
        self.held = 0

    def _run(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            slot = None
            try:
                if item[0] == "layout":
                    if self.err is None:
                        self._layout(item[1])
                else:
                    _, name, b, n_rows, text, q, slot = item
                    if self.err is None:
                        self._rows(name, b, n_rows, text, q)
            except BaseException as e:  # reported by close() on the caller's thread
                self.err = e
            finally:
                if slot is not None:
                    self.free[slot].release()


def _create_profile_files(path, row_bytes, cols, total_rows):
    """Rank 0, before the others open them: the text file and its side-car at their final size (sparse until the ranks
    have written their rows) under .partial names -- a job that dies half way leaves nothing that looks like a profile
    (the table file goes the same way); files and the side-car description of an earlier run go first."""
    for stale in (f"{path}.q6.json", path, f"{path}.q6"):
        if os.path.exists(stale):
            os.remove(stale)
    for p, size in ((path, total_rows * row_bytes), (f"{path}.q6", total_rows * cols * 4)):
        with open(p + ".partial", "wb") as f:
            f.truncate(size)


def _finish_profile_files(path, cols, total_rows):
    """Rank 0, after the barrier that says every rank's rows are written: the files get their names, then the
    side-car's description (the last thing stage 3_1 looks for)."""
    import json
    os.replace(path + ".partial", path)
    os.replace(f"{path}.q6.partial", f"{path}.q6")
    with open(f"{path}.q6.json", "w") as f:
        json.dump({"cols": int(cols) if total_rows else None, "rows": int(total_rows), "text_bytes": os.path.getsize(path)}, f)


PARSE_CHUNK_BYTES = int(os.environ.get("LRB_PARSE_CHUNK_BYTES", 1 << 26))   # (tests: many ranges of a small file)


def _rank_batches(reads_path, rank, world, threads, chunk_bytes, batch_reads, batch_bytes, packed=False):
    """(b, seqs, offs, hp) of this rank's batches; b is the batch's position in the file, hp the same batch packed by the
    parser pool (device.HostPacked) when ``packed`` is asked for and the pool parses the file, else None.
    Plain FASTA: the file is cut into byte ranges and rank r parses ranges r, r+P, ... with
    its own pool of parser threads (nobody reads what it does not own).  gzip / FASTQ
    cannot be cut: every rank streams the file and keeps every P-th batch."""
    from . import device as lrb
    from ._lib import LrbError
    serial = os.environ.get("LRB_SERIAL_READER", "0") == "1"
    if not serial:
        with lrb.ParallelReader(reads_path, threads=min(32, max(1, int(threads))), chunk_bytes=chunk_bytes,
                                rank=rank, world=world, packed=packed) as rd:
            if rd.parallel:
                try:
                    while True:
                        if packed:
                            hp = rd.next_packed()
                            if hp is None:
                                return
                            yield rd.last_range, None, hp.offs, hp
                            continue
                        batch = rd.next_batch(copy=False)
                        if batch is None:
                            return
                        yield rd.last_range, batch[0], batch[1], None
                except LrbError as e:
                    if e.code != 6:
                        raise
                    raise RuntimeError(f"{reads_path}: FASTA with '+' lines cannot be cut into ranges; "
                                       "set LRB_SERIAL_READER=1") from e
    with lrb.FastxReader(reads_path) as rd:
        b = 0
        while True:
            batch = rd.next_batch(batch_reads, batch_bytes)
            if batch is None:
                return
            if b % world == rank:
                yield b, batch[0], batch[1], None
            b += 1


def _all_counts(local, group=None):
    """{batch: reads} of every rank's batches merged (one all_gather of small dicts)."""
    rank, world = world_info(group)
    if world == 1:
        return dict(local)
    parts = [None] * world
    _dist().all_gather_object(parts, local, group=group)
    merged = {}
    for p in parts:
        merged.update(p)
    return merged


def profile_file_sharded(reads_path, output, k, bin_size, bins, threads, compute, group=None,
                         batch_reads=1 << 16, batch_bytes=1 << 28, write_table=True,
                         chunk_bytes=PARSE_CHUNK_BYTES, stats=None):
    """File-level sharded profile: writes {output}/profiles/com_profs, cov_profs (+ their .q6 side-cars) and
    (rank 0, optional) 15mers-counts exactly as the single-GPU runners do.

    Phase A parses this rank's share once, leaves every batch packed in HBM (while the budget allows), formats
    its composition rows and adds it to the rank's tallies.  One all_gather of the batches' read counts then
    gives every batch its first row in the whole file: the profile files are created at their final size by rank
    0 and EVERY rank writes its own rows at their place (fixed-width rows; ``_ShardWriter``) -- composition rows
    that were waiting in memory first, coverage rows as phase B makes them.  After the one all-reduce, phase B
    runs the coverage kernel on the batches still resident and re-parses only what did not fit.  The table file is
    shared work too: every rank holds the whole table after the all-reduce and writes 1 / world of the file at its
    place, beside phase B (lrb_k15_write_file_part_async).  Rank 0 alone only creates the files, describes the
    side-cars and gives the table file its name.  ``stats`` (dict) receives this rank's stage stamps in seconds."""
    import time
    from . import device as lrb
    dist = _dist()
    rank, world = world_info(group)
    if world > 1 and os.environ.get("LRB_DIST_FAIL_RANK") == str(rank):
        raise SystemExit(3)   # tests: a rank that dies before the first collective (tests/test_gpu_multi.py)
    os.makedirs(f"{output}/profiles", exist_ok=True)
    com_path, cov_path = f"{output}/profiles/com_profs", f"{output}/profiles/cov_profs"
    dim = lrb.kmer_dim(k)
    t_start = time.perf_counter()
    stamps = stats if stats is not None else {}

    def lap(name, t0):
        stamps[name] = round(stamps.get(name, 0.0) + time.perf_counter() - t0, 6)

    def my_batches(packed=False):
        return _rank_batches(reads_path, rank, world, threads, chunk_bytes, batch_reads, batch_bytes, packed=packed)

    writer = _ShardWriter(rank)
    writer.add("com", com_path, int(lrb.lib().lrb_com_row_bytes(dim)), dim)
    writer.add("cov", cov_path, int(lrb.lib().lrb_cov_row_bytes(int(bins))), int(bins))
    can_pack = hasattr(compute, "pack")
    # resident batches are tallied together after the loop: a group shares one pass over the table
    can_group = hasattr(compute, "k15_accumulate_many")
    budget = compute.resident_budget() if can_pack else 0
    resident, resident_bytes = {}, 0
    # phase A.  With a compute object that tallies into the CANONICAL HALF of the table (HipCompute: K2 from slice
    # lists) the rank's tallies are born folded: the half is what the ranks all-reduce, and the lists are kept for
    # phase B while memory allows; the forward-table form (fold / mirror after the tally) is what the CPU stand-ins
    # of the tests run
    half_path = hasattr(compute, "k15_tally_half_many")
    table = None if half_path else compute.new_table()
    half = compute.new_half() if half_path else None
    kept = None
    counts = {}
    table_job = None
    try:
        # (batches arrive packed by the parser pool while they can stay resident: the ASCII view is only needed by the
        # fall-backs for what does not fit, and a pool that packs hands out no ASCII)
        from . import runners_utils as _ru
        host_packs = bool(can_pack and getattr(compute, "host_packs", False) and _ru.host_packs())
        batches = my_batches(packed=host_packs)
        for b, seqs, offs, hp in batches:
            lens = np.diff(offs).astype(np.uint32)
            packed = None
            slot = None
            if can_pack and resident_bytes < budget:
                packed = compute.pack(seqs, offs, k, hp=hp) if hp is not None else compute.pack(seqs, offs, k)
                if hasattr(packed, "kmer_text"):
                    slot = writer.slot()
                    com_text, com_q = packed.kmer_text(k, slot=slot)
                else:
                    com_text, vals = lrb.format_com(packed.kmer_counts(k), lens, k, threads=threads, want_values=True)
                    com_q = _q6_of_values(vals)
                if not can_group:
                    packed.k15_accumulate(table)
            elif hp is not None:
                # past the residency budget, from a pool that hands out packed batches: the batch is packed, tallied and
                # given back (phase B parses its range again)
                tmp = compute.pack(seqs, offs, k, hp=hp)
                try:
                    slot = writer.slot()
                    com_text, com_q = tmp.kmer_text(k, slot=slot)
                    compute.k15_tally_half_many([tmp], half)
                    compute.sync()
                finally:
                    tmp.free()
            else:
                com_text, vals = lrb.format_com(compute.kmer_counts(seqs, offs, k), lens, k, threads=threads,
                                                want_values=True)
                com_q = _q6_of_values(vals)
                if half_path:
                    compute.k15_tally_half_one(seqs, offs, half)
                else:
                    compute.k15_accumulate(seqs, offs, table)
            counts[b] = len(lens)
            writer.put("com", b, len(lens), com_text, com_q, slot)
            if packed is not None:
                resident[b] = packed
                resident_bytes += packed.device_bytes
        lap("parse_pack_k1_s", t_start)
        # every batch's first row: the read counts of all ranks' batches, in file order
        t0 = time.perf_counter()
        counts_all = _all_counts(counts, group)
        first_row, total_rows = {}, 0
        for b in sorted(counts_all):
            first_row[b] = total_rows
            total_rows += counts_all[b]
        n_batches = (max(counts_all) + 1) if counts_all else 0
        table_path = f"{output}/profiles/15mers-counts"
        shared_table = bool(write_table and hasattr(compute, "ctx"))
        if rank == 0:
            _create_profile_files(com_path, writer.prof["com"]["row_bytes"], dim, total_rows)
            _create_profile_files(cov_path, writer.prof["cov"]["row_bytes"], int(bins), total_rows)
            if shared_table:
                # the table file at its full size under its .partial name: every rank writes a slice of it once the
                # table exists (they all hold it after the all-reduce); it gets its name when all slices are in
                for stale in (table_path, table_path + ".partial"):
                    if os.path.exists(stale):
                        os.remove(stale)
                with open(table_path + ".partial", "wb") as f:
                    f.truncate(8 + 4 * lrb.K15_ENTRIES)
        if world > 1:
            dist.barrier(group=group)   # the files exist
        lap("layout_s", t0)
        writer.set_layout(first_row)
        t0 = time.perf_counter()
        if half_path and resident:
            sweep_ok = 1 <= int(bins) <= 256 and os.environ.get("LRB_K3_SWEEP", "1") != "0"
            kept = compute.k15_tally_half_many(list(resident.values()), half, keep_bins=bins if sweep_ok else None)
        elif can_group and resident:
            compute.k15_accumulate_many(list(resident.values()), table)
        if stats is not None:
            compute.sync()
        lap("k2_s", t0)
        # the one collective of the path
        t0 = time.perf_counter()
        if half_path:
            allreduce_table(half, group, compute)
            table = compute.table_from_half(half)
            del half
        else:
            reduce_and_mirror(table, compute, group)
        if stats is not None:
            compute.sync()
        lap("allreduce_expand_s", t0)
        if shared_table:
            # this rank's slice, on the library's own thread and stream, beside phase B (which only reads the table)
            compute.sync()
            table_job = compute.ctx.k15_write_file_part_async(table.data_ptr(), table_path + ".partial", rank, world)
        # phase B
        t0 = time.perf_counter()

        def write_cov(b, hist, sums):
            txt, vals = lrb.format_cov(hist, sums, threads=threads, want_values=True)
            writer.put("cov", b, len(sums), txt, _q6_of_values(vals))

        if resident and hasattr(compute, "cov_text_groups") and 1 <= int(bins) <= 256 and os.environ.get("LRB_K3_SWEEP", "1") != "0":
            groups = compute.cov_text_groups(list(resident.items()), table, bin_size, bins, kept=kept, slot=writer.slot) if half_path else \
                compute.cov_text_groups(list(resident.items()), table, bin_size, bins, slot=writer.slot)
            for b, txt, q, slot in groups:
                writer.put("cov", b, counts[b], txt, q, slot)
            for packed in resident.values():
                packed.free()
        else:
            for wl in (kept or {}).values():
                wl.free()
            for b, packed in resident.items():
                if hasattr(packed, "cov_text"):
                    slot = writer.slot()
                    txt, q = packed.cov_text(table, bin_size, bins, slot=slot)
                    writer.put("cov", b, counts[b], txt, q, slot)
                else:
                    write_cov(b, *packed.cov_hist(table, bin_size, bins))
                packed.free()
        if not can_pack or resident_bytes >= budget:
            for b, seqs, offs, _ in my_batches():
                if b not in resident:
                    write_cov(b, *compute.cov_hist(seqs, offs, table, bin_size, bins))
        resident.clear()
        if stats is not None:
            compute.sync()
        lap("k3_s", t0)
        t0 = time.perf_counter()
    except BaseException:
        # the library's thread may still be copying out of `table`: it has to finish before the tensor can go (its own
        # error, if any, stays behind the one in flight); nobody gives the unfinished table file its name
        if table_job is not None:
            try:
                lrb.Context.job_wait(table_job)
            except Exception:  # noqa: BLE001
                pass
            table_job = None
        try:
            writer.close()
        except Exception:  # noqa: BLE001
            pass
        raise
    else:
        try:
            writer.close()      # this rank's rows are in the files
        except BaseException:
            if table_job is not None:
                try:
                    lrb.Context.job_wait(table_job)
                except Exception:  # noqa: BLE001
                    pass
            raise
    lap("rows_written_after_last_kernel_s", t0)
    t0 = time.perf_counter()
    if table_job is not None:
        lrb.Context.job_wait(table_job)   # this rank's 1 / world of the table file
    lap("table_file_wait_s", t0)
    t0 = time.perf_counter()
    if world > 1:
        dist.barrier(group=group)
    lap("barrier_wait_s", t0)
    t0 = time.perf_counter()
    if rank == 0:
        _finish_profile_files(com_path, dim, total_rows)
        _finish_profile_files(cov_path, int(bins), total_rows)
        if shared_table:
            os.replace(table_path + ".partial", table_path)   # every slice is in: the file gets its name
    lap("rank0_sidecar_s", t0)
    stamps["total_s"] = round(time.perf_counter() - t_start, 6)
    stamps.update({k_: v for k_, v in writer.stats.items()})
    stamps.update(rank=rank, world=world, rows=total_rows, batches=n_batches)
    sp = os.environ.get("LRB_DIST_STATS")
    if sp:
        import json
        with open(f"{sp}.rank{rank}.json", "w") as f:
            json.dump(stamps, f, indent=1)
    return n_batches


def launcher_world():
    """(rank, world, local rank) a launcher (torch.distributed.run) gave this process; (0, 1, 0) without one."""
    if "RANK" not in os.environ:
        return 0, 1, 0
    return (int(os.environ["RANK"]), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0")))


def collective_timeout():
    """How long a collective may wait for a rank before the job fails (LRB_COLLECTIVE_TIMEOUT_S, default 600 s; the
    default of torch.distributed is 10-30 min).  The collectives themselves are short (the 2 GiB all-reduce of the half
    table takes well under a second over xGMI) and no rank does host work on the others' behalf any more (every rank
    writes its own rows, rank 0's table file is written beside phase B and waited for by rank 0 alone); what a collective
    does absorb is the ranks' parse skew at the first one -- documented in ``lrbinner.py --help``, logged at start-up."""
    import datetime
    return datetime.timedelta(seconds=float(os.environ.get("LRB_COLLECTIVE_TIMEOUT_S", "600")))


def init_group():
    """The process group of a launched job: backend nccl (= RCCL over xGMI), one rank per GPU; returns the device
    index of this rank.  Rehearsal hook: LRB_DIST_BACKEND=gloo puts several ranks on ONE GPU (RCCL refuses two
    ranks on a device), so that the multi-rank path -- shards, fold, all-reduce, expand, rows written in place -- runs on the
    HIP kernels where a single MI355X is all there is (tests/test_gpu_pipeline.py)."""
    import torch
    rank, world, local = launcher_world()
    backend = os.environ.get("LRB_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local %= max(torch.cuda.device_count(), 1)
    if world > 1 and not _dist().is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # host-side stages next to the rank's GPU: before the process group's (and RCCL's) threads exist -- affinity is
        # per thread and inherited at creation
        from . import _gpus
        _gpus.pin_to_gpu_numa(local)
        # a rank that dies alone leaves the others in their next collective: minutes, not the launcher's half hour
        timeout = collective_timeout()
        if rank == 0:
            import logging
            logging.getLogger('LRBinner').info(f"{world} ranks, backend {backend}, collective timeout "
                                               f"{timeout.total_seconds():.0f} s (LRB_COLLECTIVE_TIMEOUT_S)")
        if backend == "nccl":
            _dist().init_process_group("nccl", device_id=torch.device("cuda", local), timeout=timeout)
        else:
            _dist().init_process_group(backend, timeout=timeout)
    return local


def close_group():
    if _dist().is_available() and _dist().is_initialized():
        destroy_abi_comms()
        _dist().destroy_process_group()


def spawn_ranks(n_gpus, module_args, module="lrbinner_amd.dist"):
    """Start ``python -m torch.distributed.run --nproc-per-node n_gpus -m <module> <args>`` as a CHILD job
    (rendezvous on 127.0.0.1, a free port) and return its exit status.  A new process, never an exec of the
    calling one; callers use it before they touch the GPU themselves."""
    import subprocess
    import sys
    from . import _gpus
    have = _gpus.visible_gpus()
    if have is not None and have < int(n_gpus) and os.environ.get("LRB_DIST_BACKEND", "nccl") == "nccl":
        # N ranks on fewer GPUs do not fail, they hang in the rendezvous: refuse here, with a message
        sys.stderr.write(f"lrbinner: {n_gpus} ranks asked for but this node shows {have} GPU(s)\n")
        return 2
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # --standalone: torchrun's own c10d store on a free port of 127.0.0.1 (a port picked here by bind-then-close could
    # be taken by another process before the child binds it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_gpus)}",
           "--standalone", "--local-addr", "127.0.0.1", "--master-addr", "127.0.0.1", "-m", module] + [str(a) for a in module_args]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // int(n_gpus))))
    env["PYTHONPATH"] = root + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
    for k in ("LRB_GPUS",):
        env.pop(k, None)
    return subprocess.run(cmd, env=env).returncode


def main(argv=None):
    """torchrun entry: python -m torch.distributed.run --nproc-per-node N -m lrbinner_amd.dist
    --reads R --output O [-k 3 -bs 10 -bc 32 -t 8]"""
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", required=True)
    ap.add_argument("--output", required=True)
    ap.add_argument("-k", type=int, default=3)
    ap.add_argument("-bs", type=int, default=10)
    ap.add_argument("-bc", type=int, default=32)
    ap.add_argument("-t", type=int, default=8)
    ap.add_argument("--no-table-file", action="store_true")
    a = ap.parse_args(argv)
    local = init_group()
    profile_file_sharded(a.reads, a.output, a.k, a.bs, a.bc, a.t, HipCompute(local),
                         write_table=not a.no_table_file)
    close_group()


if __name__ == "__main__":
    main()
