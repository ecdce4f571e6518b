#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for args in "3000 2000 300" "6000 2000 1563" "6000 2000 2032" "400000 10000"; do
  echo "== wl_debug $args"; timeout 900 python3 scripts/wl_debug.py $args 2>&1 | tail -12
done > gpurun_out/r04_debug.log 2>&1
cat gpurun_out/r04_debug.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -v -k "slice_lists or k3_sweep or from_slice" > gpurun_out/r04_tests_full.log 2>&1
grep -v "^  File\|^Extension" gpurun_out/r04_tests_full.log | tail -40
