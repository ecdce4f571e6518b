#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "slice_lists or k3_sweep or from_slice" > gpurun_out/r04_tests_full.log 2>&1
grep -v "^  File\|^Extension" gpurun_out/r04_tests_full.log | tail -8
CFGS="p4: p2:LRB_WL_PART_UNITS=2" bash scripts/r04_time.sh
