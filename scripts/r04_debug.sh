#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "slice_lists or k3_sweep or from_slice or c4_rank" > gpurun_out/r04_tests_full.log 2>&1
grep -E "passed|failed|error" gpurun_out/r04_tests_full.log | head -3
CFGS="dpp:" timeout 600 bash scripts/r04_time.sh 2>&1 | grep "wl_\|rc="
