"""The N>1 path of lrbinner_amd.dist under gloo, world_size 2, on CPU: read sharding,
the single all-reduce of the table (with uint32 wrap-around), mirror after the
reduce, and order-preserving output.  A 7-mer miniature of the table stands in for the
GPU (tests/dist_worker.py); the GPU kernels themselves are covered by the -m gpu tests."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from helpers import ROOT, golden_path
from lrbinner_amd import dist as ld


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, *args):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), *args]
    subprocess.run(cmd, check=True, env=env, cwd=ROOT, timeout=600)


def test_shard_ranges_tile_the_input():
    for n in (0, 1, 7, 100, 1_000_003):
        for world in (1, 2, 3, 8):
            r = [ld.shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1


def test_array_level_two_ranks_equal_one_rank(tmp_path):
    reads = golden_path("edge.fasta")
    _run(1, "array", reads, str(tmp_path / "w1.npz"))
    _run(2, "array", reads, str(tmp_path / "w2.npz"))        # fold -> all-reduce of the canonical half -> expand
    _run(2, "array_full", reads, str(tmp_path / "w2f.npz"))  # all-reduce of the whole table -> mirror
    a, b, c = np.load(tmp_path / "w1.npz"), np.load(tmp_path / "w2.npz"), np.load(tmp_path / "w2f.npz")
    assert int(a["world"]) == 1 and int(b["world"]) == 2 and int(c["world"]) == 2
    for key in ("counts", "hist", "sums"):
        assert np.array_equal(a[key], b[key]), key
        assert np.array_equal(a[key], c[key]), key
    assert a["hist"].sum() > 0


@pytest.mark.parametrize("reads,mode", [("edge.fastq", "file"), ("edge.fasta", "file"),
                                        ("edge.fasta", "file_keep"), ("edge.fasta", "file_spill")])
def test_file_level_two_ranks_equal_one_rank(tmp_path, reads, mode):
    """FASTQ: serial reader, batch-cyclic.  FASTA: byte ranges, range-cyclic through the
    library's sharded parser pool; with and without batches kept between the phases."""
    reads = golden_path(reads)
    o1, o2 = str(tmp_path / "o1"), str(tmp_path / "o2")
    _run(1, mode, reads, o1)
    _run(2, mode, reads, o2)
    assert int(open(os.path.join(o2, "nbatches")).read()) > 4
    for f in ("com_profs", "cov_profs"):
        x = open(os.path.join(o1, "profiles", f), "rb").read()
        y = open(os.path.join(o2, "profiles", f), "rb").read()
        assert x == y and len(x) > 0
    assert not [f for f in os.listdir(os.path.join(o2, "profiles")) if ".part" in f]
    # the stitched value side-car (stage 3_1 reads it instead of parsing the text) holds what the text parses to
    from lrbinner_amd.runners_utils import load_value_sidecar
    for f in ("com_profs", "cov_profs"):
        path = os.path.join(o2, "profiles", f)
        side = load_value_sidecar(path)
        assert side is not None, f
        rows = [l.split() for l in open(path).read().splitlines()]
        parsed = np.array([[float(t) for t in r] for r in rows], dtype=np.float64).reshape(len(rows), -1)
        assert side.shape == parsed.shape and np.array_equal(side, parsed), f
    # the composition text is also what the reference wrote for this input
    from helpers import gz_bytes
    assert open(os.path.join(o2, "profiles", "com_profs"), "rb").read() == gz_bytes("com_profs_k3.txt.gz")


def _run_env(world, env_extra, *args):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), *args]
    subprocess.run(cmd, check=True, env=env, cwd=ROOT, timeout=600)


def test_rows_written_in_place_with_an_empty_range_and_a_spilling_buffer(tmp_path):
    """Every rank writes its rows at their final offsets (no part files, no stitch by rank 0): a file whose byte
    ranges include some that hold NO record start (a 6 kb read across 700-byte ranges), with the rows that arrive
    before the layout is known held in memory (default) and spilled to a rank-local file (LRB_DIST_BUFFER_GB=0)."""
    import json
    rng = np.random.default_rng(5)
    recs = []
    for i in range(90):
        n = 6000 if i in (7, 50) else int(rng.integers(0, 300))
        recs.append(b">r%d\n" % i + bytes(rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=n, p=[.24, .24, .24, .24, .04])) + b"\n")
    fa = tmp_path / "reads.fasta"
    fa.write_bytes(b"".join(recs))
    outs = {}
    for name, world, env in (("w1", 1, {}), ("w2", 2, {}), ("w2spill", 2, {"LRB_DIST_BUFFER_GB": "0"}),
                             ("w3", 3, {})):
        o = str(tmp_path / name)
        _run_env(world, dict(env, LRB_DIST_STATS=str(tmp_path / (name + "_stats"))), "file_keep", str(fa), o)
        outs[name] = o
        assert not [f for f in os.listdir(os.path.join(o, "profiles")) if ".part" in f or ".spill" in f]
        st = [json.load(open(tmp_path / f"{name}_stats.rank{r}.json")) for r in range(world)]
        assert all(s["rows"] == 90 and s["world"] == world for s in st)
        if name == "w2spill":
            assert sum(s["spilled_bytes"] for s in st) > 0 and sum(s["buffered_bytes"] for s in st) == 0
        elif world > 1:
            assert sum(s["buffered_bytes"] for s in st) > 0 and sum(s["spilled_bytes"] for s in st) == 0
    for f in ("com_profs", "cov_profs", "com_profs.q6", "cov_profs.q6", "com_profs.q6.json", "cov_profs.q6.json"):
        ref = open(os.path.join(outs["w1"], "profiles", f), "rb").read()
        assert len(ref) > 0
        for name in ("w2", "w2spill", "w3"):
            assert open(os.path.join(outs[name], "profiles", f), "rb").read() == ref, (name, f)
    assert len(open(os.path.join(outs["w1"], "profiles", "com_profs")).read().splitlines()) == 90


def test_shard_writer_unit(tmp_path):
    """_ShardWriter alone: rows before the layout (memory, then spill), rows after it (direct), the staging-slot
    protocol, and a batch whose text is not n_rows x row_bytes bytes is an error reported by close()."""
    def make(cap):
        w = ld._ShardWriter(0, buffer_bytes=cap)
        path = str(tmp_path / f"p{cap}")
        w.add("x", path, 5, 2)
        return w, path

    rows = {b: (bytes([65 + b]) * 4 + b"\n") * n for b, n in ((0, 3), (1, 0), (2, 2), (5, 4))}
    qs = {b: np.full((len(t) // 5, 2), b, dtype=np.uint32) for b, t in rows.items()}
    first, total = {}, 0
    for b in sorted(rows):
        first[b] = total
        total += len(rows[b]) // 5
    want_text = b"".join(rows[b] for b in sorted(rows))
    want_q = np.concatenate([qs[b] for b in sorted(rows)]).tobytes()
    for cap in (1 << 20, 20, 0):
        w, path = make(cap)
        for b in (5, 0):            # before the layout, out of order
            s = w.slot()
            w.put("x", b, len(rows[b]) // 5, np.frombuffer(rows[b], np.uint8), qs[b], s)
        ld._create_profile_files(path, 5, 2, total)
        w.set_layout(first)
        for b in (2, 1):            # after it
            w.put("x", b, len(rows[b]) // 5, rows[b], qs[b])
        w.close()
        # the files carry their .partial names until rank 0 says every rank's rows are in (round 6)
        assert not os.path.exists(path) and not os.path.exists(path + ".q6") and not os.path.exists(path + ".q6.json")
        ld._finish_profile_files(path, 2, total)
        assert open(path, "rb").read() == want_text and open(path + ".q6", "rb").read() == want_q, cap
        assert not os.path.exists(path + ".partial") and os.path.exists(path + ".q6.json")
        assert not os.path.exists(f"{path}.rank0.spill")
        assert (w.stats["spilled_bytes"] > 0) == (cap < 1 << 20)
    w, path = make(1 << 20)
    ld._create_profile_files(path, 5, 2, 4)
    w.set_layout({0: 0})
    w.put("x", 0, 4, b"ABCD\n" * 3, np.zeros((4, 2), np.uint32))
    with pytest.raises(ValueError):
        w.close()
