#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "order_kernel_lists_per_workgroup or longer_than_the_register" > gpurun_out/r04_dbg_tests.log 2>&1
grep -n "passed\|failed\|^FAILED\|^ERROR\|^E  " gpurun_out/r04_dbg_tests.log | head -20 | cut -c1-200
