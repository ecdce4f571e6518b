#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "longer_than_the_register or c4_rank" > gpurun_out/r04_dbg_tests.log 2>&1
grep -n "passed\|failed\|rror\|assert" gpurun_out/r04_dbg_tests.log | head -20 | cut -c1-250
LRB_WL_ORDER_OCC=2 timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "longer_than_the_register or slice_lists_ragged" > gpurun_out/r04_dbg_tests2.log 2>&1
grep -n "passed\|failed\|rror\|assert" gpurun_out/r04_dbg_tests2.log | head -20 | cut -c1-250
