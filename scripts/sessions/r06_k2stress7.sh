#!/bin/bash
# does the explicit lgkmcnt(0) in front of the count kernel's barrier remove the fault (partitions repeated -> 0)?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp STRESS_TEXT=0
mkdir -p gpurun_out
for rep in 1 2 3; do
  for i in 0 1 2 3 4 5 6 7; do STRESS_TAG=$i timeout 1200 python3 scripts/k2_stress.py 80 20000 > gpurun_out/k2stress7_$i.log 2>&1 & done
  wait
  cat gpurun_out/k2stress7_*.log | grep -v amdgpu.ids | grep -E "passes|short" | cut -c1-260
done | tee gpurun_out/r06_k2_stress_lgkm.txt
rm -f gpurun_out/k2stress7_*.log
