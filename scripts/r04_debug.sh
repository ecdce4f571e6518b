#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "slice_lists or k3_sweep or from_slice" > gpurun_out/r04_tests_full.log 2>&1
grep -v "^  File\|^Extension" gpurun_out/r04_tests_full.log | tail -15
for w in 22 24; do
  LRB_WL_SWEEP_WAVES=$w timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "from_slice_lists_ragged or lists_at_size" 2>&1 | tail -3
done
CFGS="s44:LRB_WL_SWEEP_WAVES=44 s24:LRB_WL_SWEEP_WAVES=24 s22:LRB_WL_SWEEP_WAVES=22,LRB_WL_ORDER_OCC=2" bash scripts/r04_time.sh
