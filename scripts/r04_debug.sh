#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
cp ab/liblrb_new.so lrbinner_amd/liblrb_hip.so
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "lists or sweep or slice or c3_full or k2_k3 or c4_rank or order_kernel" > gpurun_out/r04_dbg_tests.log 2>&1
grep -n "passed\|failed\|^FAILED\|^ERROR" gpurun_out/r04_dbg_tests.log | head -20 | cut -c1-200
PAT="part" bash scripts/r04_ab.sh
PAT="part|order_kernel_occ1" bash scripts/r04_ab.sh | grep -v rc=
