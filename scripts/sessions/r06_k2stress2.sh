#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for i in 0 1 2 3 4 5 6 7; do STRESS_TAG=$i timeout 1200 python3 scripts/k2_stress2.py 30 20000 > gpurun_out/k2stress2_$i.log 2>&1 & done
wait
cat gpurun_out/k2stress2_*.log | grep -v "amdgpu.ids" | tail -50 | cut -c1-300
