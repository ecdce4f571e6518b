#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
LRB_WL_SWEEP_BPS=2 timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "lists or sweep or slice or c3_full or k2_k3 or c4_rank or order_kernel" > gpurun_out/r04_dbg_tests.log 2>&1
grep -n "passed\|failed\|^FAILED\|^ERROR" gpurun_out/r04_dbg_tests.log | head -20 | cut -c1-200
CFGS="b1: b2:LRB_WL_SWEEP_BPS=2 b1: b2:LRB_WL_SWEEP_BPS=2 nodb:LRB_WL_SWEEP_DB=0" bash scripts/r04_time.sh 2>&1 | grep -E "rc=|sweep" | cut -c1-110
