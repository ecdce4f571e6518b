#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== (alone: skipped)"
echo "== eight side by side"
for i in 0 1 2 3 4 5 6 7; do STRESS_TAG=$i timeout 1500 python3 scripts/k2_stress3.py 40 20000 > gpurun_out/k2stress5_$i.log 2>&1 & done
wait
cat gpurun_out/k2stress5_*.log | grep -v amdgpu.ids | tail -60 | cut -c1-700
