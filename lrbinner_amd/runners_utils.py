"""Drop-in for the profile half of ``mbcclr_utils/runners_utils.py``.

Same function names, arguments, on-disk outputs and error behaviour as the
reference (runners_utils.py:16-113); the three ``os.system`` calls to
count-kmers / count-15mers / search-15mers are replaced by calls into
liblrb_hip.so through ``lrbinner_amd.device``:

    run_kmers(reads_path, output, k_size, threads)            -> {output}/profiles/com_profs
    run_15mer_counts(reads_path, output, threads)             -> {output}/profiles/15mers-counts
    run_15mer_vecs(reads_path, output, bin_size, bin_count, threads)
                                                              -> {output}/profiles/cov_profs

``threads`` keeps its meaning for the host side only (text formatting workers);
the per-base work runs on the GPU.  Output files are truncated at the start and
rows are appended in input order, as the reference binaries do
(count-kmers.cpp:101,210; search-15mers.cpp:32,144).
"""
import logging
import atexit
import os
import pickle
import sys
from collections import defaultdict

import numpy as np

from . import device
from ._lib import LrbError

logger = logging.getLogger('LRBinner')

# one GPU batch: at most this many reads / bases (the reference uses 10,000-read
# batches, count-kmers.cpp:141; a larger batch keeps 256 CUs busy)
BATCH_READS = 1 << 17
BATCH_BYTES = 1 << 29
# byte range one parser thread turns into one batch: small enough that the pool's recycled
# buffers (threads + 2 of them) are the only memory first-touched, large enough (>= 3e7
# 15-mers) for the partitioned K2 path
PARSE_CHUNK_BYTES = 1 << 26
# K3 as a sweep (lrb_packed_cov_hist_many): resident batches are tallied this many bases at a time (16 GB of slice
# lists per 4e9); below SWEEP_MIN_BASES the per-batch gather kernel is the faster one
SWEEP_GROUP_BASES = int(os.environ.get("LRB_K2_GROUP_BASES", 4_000_000_000))   # (the library reads the same switch)
SWEEP_MIN_BASES = int(os.environ.get("LRB_K3_SWEEP_MIN_BASES", 150_000_000))
MAX_PARSER_THREADS = 32
# rows of cov_profs formatted and copied out per call after a sweep (one call a reader batch -- ~6,700 reads -- was latency:
# 0.54 ms each, 0.4 s at C3's size)
COV_CHUNK_ROWS = 1 << 16


def parser_threads(threads):
    """Size of the parser pool for a caller that asks for `threads`: at most MAX_PARSER_THREADS, and at most half the CPUs
    the process may really use (_gpus.cpu_budget: a container that shows 256 CPUs and is given the time of 16 runs the
    composition stage fastest with 8 parser threads -- 0.70 s per 2 M reads against 0.78 with 32,
    profiles/r05_threads.txt: the pool, the uploading thread and the writers share the quota)."""
    from . import _gpus
    return max(1, min(MAX_PARSER_THREADS, int(threads), max(4, _gpus.cpu_budget() // 2)))


def host_packs():
    """Do the parser threads pack the reads (2 bits a base + mask) before the upload?  0.375 bytes a base over PCIe instead
    of 1, for a third more CPU time per base in the pool (measured: 2.75 against 3.6 GB/s a thread): worth it when the
    pool is not the bottleneck -- from 32 CPUs of budget on (_gpus.cpu_budget: the cgroup quota counts).  On a box held to
    16 CPUs the composition stage is bound by the pool either way (profiles/r05_c3_stage_calls_packed.txt).
    LRB_HOST_PACK=0 / 1 decides by hand."""
    e = os.environ.get("LRB_HOST_PACK", "auto")
    if e in ("0", "1"):
        return e == "1"
    from . import _gpus
    return _gpus.cpu_budget() >= 32
# K2 from slice lists (lrb_winlists): a group of batches below this many bases is tallied by one atomic per window
K2_LISTS_MIN_BASES = int(os.environ.get("LRB_K2_LISTS_MIN_BASES", 33_000_000))

_ctx = None
_table_cache = {}  # output dir -> (device pointer, file signature)
# reads file -> packed batches kept in HBM between the three profile stages of one run
# (the reference parses the file once per binary; 288 GB of HBM make that unnecessary)
_resident = {}
_lengths = {}  # abspath -> (file signature, uint32 lengths of every record): outlives the packed batches
# reads file -> {"sig", "bins", "groups": [(ids of the group's batches, PackedLists)]}: the slice lists the table stage
# cut the windows into, kept for the coverage stage of the same reads while memory allows (4.4 bytes per base)
_kept_lists = {}
RESIDENT_BUDGET_BYTES = int(float(os.environ.get("LRB_RESIDENT_GB", "160")) * (1 << 30))


def _context():
    global _ctx
    if _ctx is None:
        _ctx = device.Context(int(os.environ.get("LRB_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    return _ctx


class Checkpointer():
    """Stage -> parameters log for --resume (runners_utils.py:16-50): a stage runs
    again when it is unknown or its parameters changed; logging a stage forgets every
    stage whose major number is larger."""

    def __init__(self, checkpoint_path, _load_to_resume=False):
        self.cpath = checkpoint_path
        self.completed = {}
        if _load_to_resume and os.path.isfile(self.cpath):
            with open(self.cpath, "rb") as f:
                self.completed = pickle.load(f)

    def should_run_step(self, stage, params):
        return stage not in self.completed or self.completed[stage] != params

    def log(self, stage, params):
        self.completed[stage] = params
        major = int(stage.split("_")[0])
        for s in list(self.completed.keys()):
            if int(s.split("_")[0]) > major:
                del self.completed[s]
        self._save()

    def _save(self):
        with open(self.cpath, "wb+") as f:
            pickle.dump(self.completed, f)

    def __str__(self):
        return str(self.completed)


def _fasta_records_b(path):
    """(id, sequence) of a FASTA file, id = header up to the first white space (str), the sequence as
    BYTES: the file is read in binary -- no decoding of gigabytes of bases, which is a third of the
    parse -- with the line handling of Bio.SeqIO's FASTA parser, which the reference reads contigs with
    (SimpleFastaParser: lines right-stripped and joined, then every ' ' and '\r' removed)."""
    name, parts = None, []
    opener = open
    if str(path).endswith(".gz"):
        import gzip
        opener = gzip.open
    with opener(path, "rb") as f:
        for raw in f:
            # (Bio reads in text mode with universal newlines: a lone '\r' ends a line as well)
            for line in (raw.split(b"\r") if b"\r" in raw else (raw,)):
                if line[:1] == b">":
                    if name is not None:
                        yield name, b"".join(parts).replace(b" ", b"").replace(b"\r", b"")
                    fields = line[1:].split()
                    name, parts = (fields[0].decode() if fields else ""), []
                elif name is not None:
                    parts.append(line.rstrip())
        if name is not None:
            yield name, b"".join(parts).replace(b" ", b"").replace(b"\r", b"")


_contig_cache = {}  # abs path -> (file signature, [ids], [sequences as bytes] or None)


class _NativeContigs:
    """The records of a contigs file parsed in one native pass (lrb_fasta_scan) and held in the library's memory:
    ``ids`` (list of str), ``lens`` (uint64 array), ``[i]`` -> the sequence of record i as bytes."""

    def __init__(self, path):
        from . import _lib
        import ctypes as C
        self._lib = _lib
        self._h = _lib.vp()
        _lib.call("lrb_fasta_scan", os.fsencode(path), C.byref(self._h))
        n, sp, op, np_, nop = C.c_uint64(0), _lib.u8p(), _lib.u64p(), _lib.u8p(), _lib.u64p()
        _lib.call("lrb_fasta_records_view", self._h, C.byref(n), C.byref(sp), C.byref(op), C.byref(np_), C.byref(nop))
        self.n = n.value
        self.offs = np.ctypeslib.as_array(op, shape=(self.n + 1,))
        no = np.ctypeslib.as_array(nop, shape=(self.n + 1,))
        self.lens = np.diff(self.offs)
        total, ntotal = int(self.offs[-1]), int(no[-1])
        self.seqs = np.ctypeslib.as_array(sp, shape=(max(total, 1),))
        names = np.ctypeslib.as_array(np_, shape=(max(ntotal, 1),)).tobytes()
        bounds = no.tolist()
        self.ids = [names[bounds[i]:bounds[i + 1]].decode() for i in range(self.n)]

    def __getitem__(self, i):
        return self.seqs[int(self.offs[i]):int(self.offs[i + 1])].tobytes()

    def write_fragments(self, out_path):
        """split_contigs' fragments file; returns the number of fragments of every record."""
        import ctypes as C
        counts = np.zeros(max(self.n, 1), dtype=np.uint32)
        total = C.c_uint64(0)
        self._lib.call("lrb_fasta_write_fragments", self._h, os.fsencode(out_path), C.byref(total),
                       counts.ctypes.data_as(self._lib.u32p))
        return counts[:self.n]

    def close(self):
        if self._h:
            self.seqs = self.offs = None
            self._lib.lib().lrb_fasta_records_free(self._h)
            self._h = self._lib.vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _contigs_native(path):
    """The cached native parse of the file, made now if the file fits comfortably in free memory; else None."""
    key = os.path.abspath(path)
    sig = _file_sig(path)
    hit = _contig_cache.get(key)
    if hit is not None and hit[0] == sig and isinstance(hit[2], _NativeContigs):
        return hit[2]
    try:
        free = os.sysconf("SC_AVPHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
    except (ValueError, OSError):
        free = 0
    # plain files are held once (the vectors grow by doubling: up to twice their size in flight); gzip expands ~4x
    need = sig[0] * (12 if str(path).endswith(".gz") else 3)
    if os.environ.get("LRB_CONTIGS_NATIVE", "1") == "0" or need >= free:
        return None
    cf = _NativeContigs(path)
    _contig_cache[key] = (sig, cf.ids, cf)
    return cf


def contig_lengths(path):
    """(ids, lengths) of the records of a contigs file, in file order."""
    cf = _contigs_native(path)
    if cf is not None:
        return cf.ids, cf.lens.tolist()
    ids, lens = [], []
    for cid, seq in contig_records(path):
        ids.append(cid)
        lens.append(len(seq))
    return ids, lens


def contig_records(path, want_seqs=True):
    """(id, sequence bytes) of every record of a contigs FASTA, in file order.  The contigs pipeline
    walks the file three times (lengths, fragmenting, output: pipelines.py:125-131,135-141,
    cluster_utils.py:512-530); here the first walk parses it in one native pass (lrb_fasta_scan) and keeps
    the records in memory while the file fits comfortably in what is free -- a Python loop over the lines
    otherwise, keeping the ids only -- and the later walks reuse that as long as the file has not changed.
    With want_seqs=False the sequence slot is None."""
    key = os.path.abspath(path)
    sig = _file_sig(path)
    hit = _contig_cache.get(key)
    if hit is None or hit[0] != sig:
        _contigs_native(path)
        hit = _contig_cache.get(key)
    if hit is not None and hit[0] == sig and (hit[2] is not None or not want_seqs):
        ids, seqs = hit[1], hit[2]
        for i, cid in enumerate(ids):
            yield cid, (seqs[i] if want_seqs else None)
        return
    keep_seqs = False  # the native parse was refused for want of memory: stream, keep the ids only
    ids, seqs = [], ([] if keep_seqs else None)
    complete = False
    try:
        for cid, seq in _fasta_records_b(path):
            ids.append(cid)
            if keep_seqs:
                seqs.append(seq)
            yield cid, seq
        complete = True
    finally:
        if complete:
            _contig_cache[key] = (sig, ids, seqs)


def release_contigs(path=None):
    """Forget the cached contig records (one file or all)."""
    keys = list(_contig_cache) if path is None else [os.path.abspath(path)]
    for k in keys:
        hit = _contig_cache.pop(k, None)
        if hit is not None and isinstance(hit[2], _NativeContigs):
            hit[2].close()


def _fasta_records(path):
    """(id, sequence) of a FASTA file; id = header up to the first white space."""
    for name, seq in _fasta_records_b(path):
        yield name, seq.decode()


def split_contigs(contigs, output):
    """Contigs >= 5000 bp become 2500-bp windows plus the last 2500 bp; shorter ones
    stay whole.  Writes {output}/fragments/contigs.fasta (ids ``>{n}_{i}``) and returns
    (contig id -> fragment numbers, fragment number -> contig id).
    runners_utils.py:53-75."""
    contig_groups = defaultdict(list)
    fragment_parent = {}
    cf = _contigs_native(contigs)
    if cf is not None:
        # the file is in memory: the fragments file is written by the library, the two maps follow from the counts
        counts = cf.write_fragments(f"{output}/fragments/contigs.fasta").tolist()
        i = 0
        for rid, c in zip(cf.ids, counts):
            contig_groups[rid].extend(range(i, i + c))
            i += c
        fragment_parent = dict(enumerate(np.repeat(np.array(cf.ids, dtype=object), counts).tolist()))
        return contig_groups, fragment_parent
    with open(f"{output}/fragments/contigs.fasta", "wb", buffering=1 << 22) as scf:
        i = 0
        for n, (rid, seq) in enumerate(contig_records(contigs)):
            if len(seq) >= 5000:
                pieces = [seq[x:x + 2500] for x in range(0, len(seq), 2500)]
                pieces.append(seq[-2500:])
            else:
                pieces = [seq]
            group = contig_groups[rid]
            for piece in pieces:
                scf.write(b">%d_%d\n%b\n" % (n, i, piece))
                group.append(i)
                fragment_parent[i] = rid
                i += 1
    return contig_groups, fragment_parent


class _RestartSerial(Exception):
    """The parallel reader met a file it cannot cut into ranges; redo the stage serially."""


_serial_only = set()


def _batches(reads_path, threads=8, packed=False):
    """(seqs, offs) batches of the file in order.  Plain FASTA comes from the library's
    pool of parser threads (views into library memory, valid until the next batch);
    gzip / FASTQ input and files the pool refuses come from the serial reader.
    packed=True: device.HostPacked batches instead -- the pool packs every range to 2 bits a base + validity mask in the
    thread that parsed it, so that 0.375 bytes a base cross PCIe instead of 1 (the serial reader's batches are packed
    here)."""
    key = os.path.abspath(reads_path)
    if key not in _serial_only and os.environ.get("LRB_SERIAL_READER", "0") != "1":
        # the parser pool feeds the GPU with 32 threads and only loses beyond (measured with the drop-in
        # executables: 1 M x 10 kb in 0.43 s with 32 threads, 1.0 s with 256)
        with device.ParallelReader(reads_path, threads=parser_threads(threads),
                                   chunk_bytes=PARSE_CHUNK_BYTES, packed=packed) as rd:
            while True:
                try:
                    b = rd.next_packed() if packed else rd.next_batch(copy=False)
                except LrbError as e:
                    if e.code == 6:
                        _serial_only.add(key)
                        raise _RestartSerial() from e
                    raise
                if b is None:
                    return
                yield b
    else:
        with device.FastxReader(reads_path) as rd:
            while True:
                b = rd.next_batch(BATCH_READS, BATCH_BYTES)
                if b is None:
                    return
                yield device.pack_reads_host(*b) if packed else b


def release_lists(reads_path=None):
    """Free the slice lists kept for the coverage stage of one reads file (or of all)."""
    keys = [os.path.abspath(reads_path)] if reads_path is not None else list(_kept_lists)
    for k in keys:
        ent = _kept_lists.pop(k, None)
        if ent:
            for _, wl in ent["groups"]:
                wl.free()


def _batch_groups(batches, max_bases, bases_of=lambda b: b.total_bases):
    """Consecutive batches in groups of at most max_bases bases (one batch at least), filled FROM THE END -- the rule of
    lrb_packed_group_starts, so that the groups the coverage stage forms are the ones the table stage's
    lrb_packed_k15_tally_half_many formed and the LAST one, whose slice lists that call leaves in the workspaces, is a
    full one.  Yields (group, bases) in input order."""
    batches = list(batches)
    groups, end = [], len(batches)
    while end > 0:
        start, bases = end, 0
        while start > 0 and (start == end or bases + bases_of(batches[start - 1]) <= max_bases):
            bases += bases_of(batches[start - 1])
            start -= 1
        groups.append((batches[start:end], bases))
        end = start
    yield from reversed(groups)


_releasing = []  # threads still handing resident batches back to the device (release_resident(background=True))


def _join_releases():
    while _releasing:
        _releasing.pop().join()


def release_resident(reads_path=None, background=False):
    """Free the HBM-resident batches of one reads file (or of all).  ``background``: on a thread of its own -- a batch is
    eight device allocations and a hipFree costs ~0.1 ms, so the 750 batches of a 5 M-read file take half a second to hand
    back, as long as the coverage stage's kernels; the last profile stage has no use for that memory and nothing after
    it waits for it (the next stage that makes batches, and the interpreter's exit, join the thread first)."""
    release_lists(reads_path)
    keys = [os.path.abspath(reads_path)] if reads_path is not None else list(_resident)
    batches = []
    for k in keys:
        ent = _resident.pop(k, None)
        if ent:
            batches.extend(ent["batches"])
    if background and batches:
        import threading

        def work():
            for b in batches:
                b.free()

        th = threading.Thread(target=work, daemon=True)
        th.start()
        _releasing.append(th)
        return
    _join_releases()
    for b in batches:
        b.free()


def _k1_layout(k_size):
    """The transposed layout the composition kernel of this k reads (lrb_packed_create flags)."""
    return 1 if k_size == 3 else 2


def _resident_batches(reads_path, with_planes=0, threads=8):
    """ResidentBatch objects of the whole file, in order.  Served from HBM when an
    earlier stage of this process left them there (and the file has not changed);
    otherwise parsed, uploaded, packed -- and kept while the budget allows."""
    key = os.path.abspath(reads_path)
    sig = _file_sig(reads_path) if os.path.exists(reads_path) else None
    ent = _resident.get(key)
    with_planes = int(with_planes)
    if ent and ent["complete"] and ent["sig"] == sig and (ent["planes"] & with_planes) == with_planes:
        for b in ent["batches"]:
            yield b
        return
    release_resident(reads_path)
    _join_releases()   # (batches an earlier stage is still handing back: their memory first)
    ctx = _context()
    ent = {"sig": sig, "batches": [], "complete": False, "planes": with_planes, "bytes": 0}
    # what may stay resident: the configured ceiling, and never more than 60 % of the HBM that is free
    # right now (the 4 GiB table, the partition buffers and the VAE stage need the rest; a smaller or
    # shared GPU simply streams the file again in the later stages)
    budget = RESIDENT_BUDGET_BYTES - sum(e["bytes"] for e in _resident.values())
    try:
        budget = min(budget, int(ctx.mem_info()[0] * 0.6))
    except LrbError:
        pass
    keep = budget > 0
    finished = False
    lens_seen = []

    def drop_kept():
        for old in ent["batches"]:
            old.free()
        ent["batches"], ent["bytes"] = [], 0

    try:
        hostp = host_packs()

        def make(hb):
            return ctx.packed_create_packed(hb, with_planes=with_planes) if hostp else ctx.packed_create(hb[0], hb[1], with_planes=with_planes)

        for hb in _batches(reads_path, threads, packed=hostp):
            try:
                b = make(hb)
            except LrbError as e:
                if e.code != 3 or not ent["batches"]:
                    raise
                keep = False  # LRB_ERR_NOMEM with batches held: give them back and stream from here on
                drop_kept()
                b = make(hb)
            lens_seen.append(b.lens)
            if keep and ent["bytes"] + b.device_bytes > budget:
                keep = False  # too big to stay resident: later stages re-read the file
                drop_kept()
            try:
                yield b
            finally:
                if keep:
                    ent["batches"].append(b)
                    ent["bytes"] += b.device_bytes
                else:
                    b.free()
        finished = True
    finally:
        if finished:
            _lengths[key] = (sig, np.concatenate(lens_seen) if lens_seen else np.zeros(0, np.uint32))
        if keep and finished:
            ent["complete"] = True
            _resident[key] = ent
        else:  # consumer stopped early or failed: nothing stays behind
            for old in ent["batches"]:
                old.free()


def read_lengths(reads_path, threads=8):
    """uint32 length of every record of the file, in order (lengths.txt, cluster_utils.py:343-349).
    From the profile stages of this process when they saw the same file; parsed otherwise."""
    key = os.path.abspath(reads_path)
    sig = _file_sig(reads_path) if os.path.exists(reads_path) else None
    hit = _lengths.get(key)
    if hit is not None and hit[0] == sig:
        return hit[1]
    parts = [np.diff(offs).astype(np.uint32) for _, offs in _batches(reads_path, threads)]
    lens = np.concatenate(parts) if parts else np.zeros(0, np.uint32)
    _lengths[key] = (sig, lens)
    return lens


class _ValueSidecar:
    """The six-decimal integers of a text profile (uint32, value = q / 1e6), written next to it
    while the text is being produced: ``{profile}.q6`` plus ``{profile}.q6.json`` (row width
    and the size of the text file it belongs to).  Stage 3_1 (text -> npy,
    pipelines.py:315-321) then reads these instead of re-parsing hundreds of MB of text; q / 1e6
    is the number ``float(token)`` gives for the token the text holds, bit for bit (both are the
    correctly rounded double of the same decimal)."""

    def __init__(self, text_path):
        self.text_path = text_path
        self.path = text_path + ".q6"
        self.f = open(self.path, "wb")
        self.cols = None
        self.rows = 0

    def append(self, q):
        if q.shape[0]:
            self.cols = int(q.shape[1])
            self.rows += int(q.shape[0])
            np.ascontiguousarray(q, dtype=np.uint32).tofile(self.f)

    def reserve(self, q):
        """(file descriptor, byte offset) where the rows of q belong -- for a writer that puts them there itself (pwrite);
        counted as appended."""
        self.f.flush()
        at = 4 * self.rows * (self.cols or int(q.shape[1]))
        self.cols = int(q.shape[1])
        self.rows += int(q.shape[0])
        return self.f.fileno(), at

    def close(self):
        import json
        self.f.close()
        with open(self.path + ".json", "w") as f:
            json.dump({"cols": self.cols, "rows": self.rows,
                       "text_bytes": os.path.getsize(self.text_path)}, f)


def q6_to_values(q):
    """float64 q / 1e6 of six-decimal integers (uint32 <= 10^6): the double ``float(token)`` gives
    for the token the text holds.  Through torch's thread pool when it is there (the division is
    the same correctly rounded IEEE operation; tests hold the two paths equal for every q)."""
    q = np.ascontiguousarray(q, dtype=np.uint32)
    try:
        import torch
        return torch.from_numpy(q.view(np.int32)).to(torch.float64).div_(1e6).numpy()
    except ImportError:
        vals = q.astype(np.float64)
        vals /= 1e6
        return vals


def load_value_sidecar(text_path):
    """float64 [rows, cols] from the side-car of ``text_path`` or None when it is absent
    or does not belong to the current text file."""
    import json
    try:
        with open(text_path + ".q6.json") as f:
            meta = json.load(f)
        if meta["text_bytes"] != os.path.getsize(text_path) or not meta["cols"]:
            return None
        flat = np.fromfile(text_path + ".q6", dtype=np.uint32)
        if flat.size != meta["rows"] * meta["cols"]:
            return None
        return q6_to_values(flat).reshape(meta["rows"], meta["cols"])
    except (OSError, KeyError, ValueError):
        return None


class _ProfileWriter:
    """Writes (text, q6) pairs to a profile file and its side-car from a few threads of its own, so that the ~1.3 KB per
    read of text (k = 4) goes to the page cache while the next batch is parsed, tallied and formatted.  The bytes of a
    pair are appended -- the caller's order is the file's order -- but written by pwrite at the offsets that order
    fixes, in pieces of at most 4 MB handed to WORKERS threads (a single writer thread was the composition stage's
    bottleneck: 3.5 GB of page-cache copies per 2 M reads, 0.6 s on one thread).  SLOTS staging slots: ``slot()`` hands out
    the next one once its previous contents have been written (waiting for the writers if need be); ``put`` queues what
    was formatted into it -- eight, so that the coverage stage can format a whole group's rows and launch the next group's
    sweep while the writers are still busy with the last one's.  os.pwrite releases the GIL."""
    WORKERS = 6
    SLOTS = 8
    PIECE = 4 << 20

    def __init__(self, out, side):
        import queue
        import threading
        self.out, self.side = out, side
        out.flush()
        self.fd = out.fileno()
        self.text_at = out.tell()
        self.q = queue.Queue()
        self.free = [threading.Semaphore(1) for _ in range(self.SLOTS)]
        self.pending = [0] * self.SLOTS          # pieces of a slot's pair still being written
        self.lock = threading.Lock()
        self.err = None
        self.turn = 0
        self.threads = [threading.Thread(target=self._run, daemon=True) for _ in range(self.WORKERS)]
        for th in self.threads:
            th.start()

    def _run(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            slot, fd, buf, at = item
            try:
                if self.err is None:
                    done = 0
                    while done < len(buf):
                        done += os.pwrite(fd, buf[done:], at + done)
            except BaseException as e:  # reported by close() on the caller's thread
                self.err = e
            finally:
                with self.lock:
                    self.pending[slot] -= 1
                    last = self.pending[slot] == 0
                if last:
                    self.free[slot].release()

    def slot(self):
        s = self.turn
        self.turn = (self.turn + 1) % self.SLOTS
        self.free[s].acquire()
        return s

    def put(self, slot, txt, q6):
        pieces = []
        tv = memoryview(txt).cast("B")
        for o in range(0, len(tv), self.PIECE):
            pieces.append((self.fd, tv[o:o + self.PIECE], self.text_at + o))
        self.text_at += len(tv)
        if q6 is not None and q6.shape[0]:
            fdq, at = self.side.reserve(q6)
            qv = memoryview(np.ascontiguousarray(q6, dtype=np.uint32)).cast("B")
            for o in range(0, len(qv), self.PIECE):
                pieces.append((fdq, qv[o:o + self.PIECE], at + o))
        if not pieces:
            self.free[slot].release()
            return
        with self.lock:
            self.pending[slot] = len(pieces)
        for fd, buf, at in pieces:
            self.q.put((slot, fd, buf, at))

    def seek_rows(self, row, row_bytes, cols):
        """The next pair belongs at row ``row`` of the file (fixed-width rows of row_bytes text bytes / cols values): a
        caller that formats a later group of rows FIRST (the coverage stage: the group whose slice lists are still in the
        workspaces) and comes back for the others.  The last call has to leave the position at the end of the file."""
        if not hasattr(self, "text_base"):
            self.text_base = self.text_at
            self.row_base = self.side.rows
        self.text_at = self.text_base + int(row) * int(row_bytes)
        self.side.f.flush()
        self.side.rows = self.row_base + int(row)
        self.side.cols = self.side.cols or int(cols)

    def close(self):
        for _ in self.threads:
            self.q.put(None)
        for th in self.threads:
            th.join()
        self.out.seek(self.text_at)    # (the file object's own position: the caller's flush / close follow)
        if self.err is not None:
            raise self.err


def _guard(step_name, fn):
    """Run fn(); map any failure to the reference's non-zero-exit convention."""
    try:
        try:
            fn()
        except _RestartSerial:
            release_resident()
            fn()  # once more, on the serial reader (outputs are truncated at the start)
        ret = 0
    except (LrbError, OSError, MemoryError) as e:
        logger.error(str(e))
        ret = 1
    check_proc(ret, step_name)


def run_kmers(reads_path, output, k_size, threads):
    if not os.path.isdir(f"{output}/profiles"):
        os.makedirs(f"{output}/profiles")
    out_path = f"{output}/profiles/com_profs"
    logger.debug(f"HIP::composition k={k_size} {reads_path} -> {out_path}")

    def work():
        ctx = _context()
        n = 0
        with open(out_path, "wb") as out:
            side = _ValueSidecar(out_path)
            wr = _ProfileWriter(out, side)
            try:
                for batch in _resident_batches(reads_path, with_planes=_k1_layout(k_size), threads=threads):
                    slot = wr.slot()
                    txt, q = batch.kmer_text(k_size, slot=slot)  # K1 + K8: counted and formatted in HBM
                    wr.put(slot, txt, q)
                    n += batch.n
            finally:
                wr.close()
            out.flush()
            side.close()
        logger.debug(f"composition vectors for {n} reads")

    _guard("Counting Trimers", work)


_pending_tables = {}  # output dir -> job writing {output}/profiles/15mers-counts from the cached table


def finish_table_files(output=None):
    """Wait for the table files still being written in the background (one output directory, or
    all) and give their 4 GiB of HBM back.  Raises LrbError if a write failed."""
    keys = [os.path.abspath(output)] if output is not None else list(_pending_tables)
    for key in keys:
        job = _pending_tables.pop(key, None)
        if job is None:
            continue
        try:
            device.Context.job_wait(job)
        finally:
            ent = _table_cache.pop(key, None)
            if ent is not None:
                _context().free(ent[0])


def _finish_at_exit():
    try:
        finish_table_files()
    except Exception as e:  # nothing to report to at this point but the log
        logger.error(f"15-mer table file: {e}")
    _join_releases()


atexit.register(_finish_at_exit)


def _drop_table(output):
    key = os.path.abspath(output)
    if key in _pending_tables:
        finish_table_files(output)  # the writer reads the table: it goes first
        return
    ent = _table_cache.pop(key, None)
    if ent is not None:
        _context().free(ent[0])


def _file_sig(path):
    st = os.stat(path)
    return (st.st_size, st.st_mtime_ns)


def run_15mer_counts(reads_path, output, threads, defer_table_file=False, coverage_bins=None):
    """count-15mers.  ``coverage_bins``: the histogram width of a run_15mer_vecs call that will follow on the SAME
    reads (the pipeline's bin_count): the slice lists this stage cuts the windows into are then kept for it while
    memory allows, and the coverage stage starts at its sweep."""
    if not os.path.isdir(f"{output}/profiles"):
        os.makedirs(f"{output}/profiles")
    out_path = f"{output}/profiles/15mers-counts"
    logger.debug(f"HIP::15-mer table {reads_path} -> {out_path}")

    def work():
        ctx = _context()
        _drop_table(output)
        release_lists(reads_path)
        table = ctx.alloc_table()
        half = None
        try:
            # the tallies go into the CANONICAL HALF of the table (one counter per pair x / rc(x): T[x] = T[rc(x)] =
            # H[h(x)], kmer_utils.h:139-153), from slice lists for groups of batches and by single atomics for crumbs
            half = ctx.alloc_half()
            key = os.path.abspath(reads_path)
            ent = _resident.get(key)
            sig = _file_sig(reads_path) if os.path.exists(reads_path) else None
            # Lists of their own memory, kept for the coverage stage, are for long-lived hosts that allocate once
            # (LRB_KEEP_LISTS=1; bench.py's C4 phases do the same with preallocated buffers): a 16 GB hipMalloc costs
            # 0.4 s, forty times the partition pass it saves, so a one-shot run makes its lists in the context's
            # workspaces and lets the coverage stage partition again
            keep = None
            if coverage_bins is not None and 1 <= int(coverage_bins) <= 256 and os.environ.get("LRB_K3_SWEEP", "1") != "0" \
                    and os.environ.get("LRB_KEEP_LISTS", "0") == "1":
                keep = {"sig": sig, "bins": int(coverage_bins), "groups": []}
            lists_bins = min(int(coverage_bins), 145) if keep else 32

            def tally(group, bases, may_keep):
                # kept while a third of what is free is not needed for it (the table file's staging, the VAE's
                # matrices and the partition workspaces come later)
                own = bool(may_keep and keep is not None and bases >= max(SWEEP_MIN_BASES, K2_LISTS_MIN_BASES)
                           and bases * 14 < ctx.mem_info()[0])
                if not own:
                    ctx.k15_tally_half_many(group, half)   # lists in the context's workspaces (or single atomics for crumbs)
                    return
                wl = device.PackedLists(ctx, group, lists_bins, workspace=False)
                try:
                    wl.tally(half)
                except BaseException:
                    wl.free()
                    raise
                if wl.fits(coverage_bins):
                    keep["groups"].append((tuple(id(b) for b in group), wl))
                else:
                    ctx.sync()
                    wl.free()

            if ent and ent["complete"] and ent["sig"] == sig and keep is None:
                # an earlier stage left the whole file packed in HBM: one library call forms the groups of batches
                # that share a partition of their windows (lrb_packed_k15_tally_half_many)
                # (the last group's lists stay in the workspaces for run_15mer_vecs of the same reads)
                ctx.k15_tally_half_many(ent["batches"], half, bins=coverage_bins or 32)
            elif ent and ent["complete"] and ent["sig"] == sig:
                for group, bases in _batch_groups(ent["batches"], SWEEP_GROUP_BASES):
                    tally(group, bases, True)
            else:
                for batch in _resident_batches(reads_path, threads=threads):
                    tally([batch], batch.total_bases, False)
            ctx.k15_expand_half(half, table)
            ctx.free(half)
            half = None
            if keep is not None and keep["groups"] and _resident.get(key, {}).get("complete"):
                _kept_lists[key] = keep
            elif keep is not None:
                for _, wl in keep["groups"]:
                    wl.free()
            if defer_table_file:
                # the pipeline's own call: the 4 GiB file is written on the library's thread while the
                # coverage stage (which reads the table from HBM) and the stages after it run;
                # finish_table_files() -- run_15mer_vecs of another table, the pipeline before
                # clustering, interpreter exit -- waits for it.  The file appears under its name
                # only when complete.
                if os.path.exists(out_path):
                    os.remove(out_path)
                job = ctx.k15_write_file_async(table, out_path)
            else:
                ctx.k15_write_file(table, out_path)
        except BaseException:
            if half is not None:
                ctx.free(half)
            ctx.free(table)
            release_lists(reads_path)
            raise
        # (the partition buffers of the accumulate stay for the coverage stage: run_15mer_vecs gives them back)
        # keep the table in HBM for run_15mer_vecs of the same run
        key = os.path.abspath(output)
        if defer_table_file:
            _table_cache[key] = (table, None)
            _pending_tables[key] = job
        else:
            _table_cache[key] = (table, _file_sig(out_path))

    _guard("Counting 15-mers", work)


def run_15mer_vecs(reads_path, output, bin_size, bin_count, threads):
    if not os.path.isdir(f"{output}/profiles"):
        os.makedirs(f"{output}/profiles")
    table_path = f"{output}/profiles/15mers-counts"
    out_path = f"{output}/profiles/cov_profs"
    logger.debug(f"HIP::coverage bs={bin_size} bc={bin_count} {reads_path} -> {out_path}")

    def work():
        ctx = _context()
        key = os.path.abspath(output)
        ent = _table_cache.get(key)
        pending = key in _pending_tables  # this process is still writing that very table to the file
        if ent is not None and pending:
            table = ent[0]
        elif ent is not None and os.path.exists(table_path) and ent[1] == _file_sig(table_path):
            table = ent[0]
        else:
            _drop_table(output)
            table = ctx.alloc_table()
            try:
                ctx.k15_read_file(table, table_path)
            except BaseException:
                ctx.free(table)
                raise
            _table_cache[key] = (table, _file_sig(table_path))
        with open(out_path, "wb") as out:
            side = _ValueSidecar(out_path)
            wr = _ProfileWriter(out, side)
            cmap = None
            try:
                if 1 <= int(bin_count) <= 256 and os.environ.get("LRB_K3_SWEEP", "1") != "0":
                    # K3 as a sweep over the compact map of the table (one bin id per pair x / rc(x), 512 MB):
                    # resident batches are tallied SWEEP_READS at a time -- one reader batch is too few reads for it
                    cmap = ctx.cov_map_build(table, bin_size, bin_count)
                group, bases = [], 0
                # batches can only wait for their group while they are sure to stay alive: the generator frees a
                # streamed batch as soon as it is asked for the next one (and everything it kept when the budget runs
                # out mid-file), so a file that is not resident from the earlier stages is tallied batch by batch
                ent = _resident.get(os.path.abspath(reads_path))
                may_group = bool(ent and ent["complete"] and os.path.exists(reads_path) and ent["sig"] == _file_sig(reads_path))

                kept = _kept_lists.get(os.path.abspath(reads_path)) if may_group else None
                kept_by_group = dict(kept["groups"]) if kept and kept["sig"] == ent["sig"] else {}

                def flush():
                    nonlocal group, bases
                    if not group:
                        return
                    wl = kept_by_group.pop(tuple(id(b) for b in group), None)
                    if cmap is not None and wl is not None and wl.fits(bin_count):
                        # the table stage left this group's slice lists: K3 is the sweep alone
                        for slot, txt, q in wl.cov_text(cmap, bin_count, slot=wr.slot, chunk_rows=COV_CHUNK_ROWS):
                            wr.put(slot, txt, q)
                        ctx.sync()
                        wl.free()
                    elif cmap is not None and bases >= SWEEP_MIN_BASES:
                        for slot, txt, q in ctx.cov_text_many(group, cmap, bin_count, slot=wr.slot, chunk_rows=COV_CHUNK_ROWS):
                            wr.put(slot, txt, q)
                    else:
                        for b in group:
                            slot = wr.slot()
                            txt, q = b.cov_text(table, bin_size, bin_count, slot=slot)  # K3 + K8
                            wr.put(slot, txt, q)
                    group, bases = [], 0

                if may_group:
                    # groups as the table stage formed them (_batch_groups) so that its lists can be found again.  The
                    # LAST group's are still in the workspaces when that stage was lrb_packed_k15_tally_half_many and
                    # nothing has partitioned since: that group is swept FIRST, as it stands -- its rows go to their
                    # place further down the file -- and the others follow in order (each partitions over those lists)
                    groups = list(_batch_groups(_resident_batches(reads_path, threads=threads), SWEEP_GROUP_BASES))
                    ahead = bool(len(groups) > 1 and cmap is not None and not kept_by_group
                                 and ctx.lists_resident(groups[-1][0], bin_count))
                    row_bytes = int(device.lib().lrb_cov_row_bytes(int(bin_count)))
                    before = sum(b.n for g, _ in groups[:-1] for b in g)
                    if ahead:
                        wr.seek_rows(before, row_bytes, int(bin_count))
                        group, bases = groups[-1]
                        flush()
                        wr.seek_rows(0, row_bytes, int(bin_count))
                    for g, gb in (groups[:-1] if ahead else groups):
                        group, bases = g, gb
                        flush()
                    if ahead:
                        wr.seek_rows(before + sum(b.n for b in groups[-1][0]), row_bytes, int(bin_count))
                else:
                    for batch in _resident_batches(reads_path, threads=threads):
                        group, bases = [batch], batch.total_bases
                        flush()
            finally:
                wr.close()
                if cmap is not None:
                    ctx.free(cmap)
            out.flush()
            side.close()
        if not pending:
            _drop_table(output)  # 4 GiB of HBM back before the VAE stage
        release_resident(reads_path, background=True)  # coverage is the last profile stage of a run
        ctx.trim()   # and the partition / slice-list workspaces

    _guard("Counting 15-mer profiles", work)


def check_proc(ret, name=""):
    if ret != 0:
        if name != "":
            logger.error(f"Error in step: {name}")
        logger.error("Failed due to an error. Please check the log. Good Bye!")
        sys.exit(ret)
