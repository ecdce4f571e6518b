#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2; do
  for i in 0 1 2 3 4 5 6 7; do STRESS_TAG=$i timeout 1200 python3 scripts/k13_stress.py 60 20000 > gpurun_out/k13stress_$i.log 2>&1 & done
  wait
  cat gpurun_out/k13stress_*.log | grep -v amdgpu.ids | grep -E "passes|pass [0-9]" | cut -c1-260
done | tee gpurun_out/r06_k13_stress.txt
