#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "slice_lists or k3_sweep or from_slice" > gpurun_out/r04_tests_full.log 2>&1
grep -v "^  File\|^Extension" gpurun_out/r04_tests_full.log | tail -6
CFGS="base:" bash scripts/r04_time.sh
timeout 1500 python3 scripts/c1_hard_explore.py w6_300_900 w10_300_900 c10_300_900 w3_300_900 w6_200_1000 w15_300_900 2>&1 | grep -v amdgpu.ids | tail -12
