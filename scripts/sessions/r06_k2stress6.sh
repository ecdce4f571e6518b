#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp STRESS_TEXT=0
mkdir -p gpurun_out
for rep in 1 2 3; do
  for i in 0 1 2 3 4 5 6 7; do STRESS_TAG=$i timeout 1200 python3 scripts/k2_stress.py 80 20000 > gpurun_out/k2stress6_$i.log 2>&1 & done
  wait
  cat gpurun_out/k2stress6_*.log | grep -v amdgpu.ids | grep -E "passes|short" | cut -c1-260
done | tee gpurun_out/r06_k2_stress_after_check.txt
echo "== alone"; STRESS_TAG=9 timeout 600 python3 scripts/k2_stress.py 80 20000 2>&1 | grep passes
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "lists or slice or sweep or c3_full or c4" 2>&1 | tail -3
