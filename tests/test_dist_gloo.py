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
