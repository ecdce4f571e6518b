#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "lists or sweep or slice or c3_full or k2_k3 or c4_rank" > gpurun_out/r04_dbg_tests.log 2>&1
grep -n "passed\|failed\|rror\|assert" gpurun_out/r04_dbg_tests.log | head -20 | cut -c1-250
CFGS="a: b: big:K2K3_N=393216,LRB_K3_SWEEP_READS=1536" bash scripts/r04_time.sh 2>&1 | grep -E "rc=|order|tally|error" | cut -c1-150
