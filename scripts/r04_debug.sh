#!/bin/bash
# scratch: the command line of the round's last short GPU session (list / sweep tests, then K2 / K3 kernel times)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "lists or sweep or slice or c3_full or k2_k3 or c4_rank or order_kernel" > gpurun_out/r04_dbg_tests.log 2>&1
grep -n "passed\|failed\|^FAILED\|^ERROR" gpurun_out/r04_dbg_tests.log | head -20 | cut -c1-200
CFGS="a: b:" bash scripts/r04_time.sh 2>&1 | grep -E "part|order_kernel_occ1|tally|count|sweep" | cut -c1-110
