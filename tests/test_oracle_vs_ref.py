"""The oracle against the REAL reference binary on fuzzed inputs (reader semantics +
composition text).  Runs wherever oracle/_ref exists (built from /root/reference by
oracle/Makefile; git-ignored, travels to the GPU box with the snapshot); skipped
otherwise."""
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as orc
from test_host_logic import _random_fastx

REF = orc.ref_bin("count-kmers")
pytestmark = pytest.mark.skipif(REF is None, reason="oracle/_ref not built (needs /root/reference)")


def test_fuzzed_files_same_text_as_reference_binary(tmp_path):
    rng = np.random.default_rng(123)
    p, out = str(tmp_path / "f.fx"), str(tmp_path / "com")
    checked = 0
    for trial in range(60):
        blob = _random_fastx(rng)
        if b"\x00" in blob:
            continue
        with open(p, "wb") as f:
            f.write(blob)
        subprocess.run([REF, p, out, "3", "2"], check=True, stdout=subprocess.DEVNULL)
        buf, offs = orc.fastx_read(p)
        counts, totals = orc.count_kmers(buf, offs, 3)
        assert orc.format_com(orc.com_profile(counts, totals)) == open(out, "rb").read(), (trial, blob)
        checked += 1
    assert checked > 40
