#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for cfg in 0 1 0 1; do
  echo "== LRB_CONCAT_KERNEL=$cfg"
  for i in 0 1 2 3 4 5 6 7; do LRB_CONCAT_KERNEL=$cfg STRESS_TAG=$i timeout 1200 python3 scripts/k2_stress2.py 40 20000 > gpurun_out/k2stress3_$i.log 2>&1 & done
  wait
  cat gpurun_out/k2stress3_*.log | grep "passes:" | cut -c1-200
done
