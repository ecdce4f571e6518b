#!/bin/bash
# scratch: the streamed order kernel for every list (LRB_WL_ORDER_OCC=2) and a one-list run length through the list tests
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
LRB_WL_ORDER_OCC=2 timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "slice_lists or k3_sweep_ragged or longer_than" > gpurun_out/r04_dbg_tests.log 2>&1
grep -n "passed\|failed\|^FAILED\|^ERROR" gpurun_out/r04_dbg_tests.log | head -5 | cut -c1-200
LRB_WL_ORDER_RUN=1 timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "slice_lists or c3_full or c4_rank" > gpurun_out/r04_dbg_tests2.log 2>&1
grep -n "passed\|failed\|^FAILED\|^ERROR" gpurun_out/r04_dbg_tests2.log | head -5 | cut -c1-200
