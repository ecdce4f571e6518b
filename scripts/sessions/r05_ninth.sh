#!/bin/bash
# after the batches' codes stopped being copied for a group's partition: list / pipeline / multi tests, the C4 gap probe, bench
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py tests/test_gpu_multi.py tests/test_gpu_bins.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | head -5
python3 scripts/c4_gap_probe.py 2500000
python3 scripts/c3_stage_probe.py 5000000 2>&1 | grep -v amdgpu.ids
