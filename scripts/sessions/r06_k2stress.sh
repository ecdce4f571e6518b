#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== eight processes side by side"
for i in 0 1 2 3 4 5 6 7; do STRESS_TAG=$i timeout 900 python3 scripts/k2_stress.py 40 20000 > gpurun_out/k2stress_$i.log 2>&1 & done
wait
cat gpurun_out/k2stress_*.log | grep -v "^$" | tail -40 | cut -c1-400
echo "== one process alone"
STRESS_TAG=9 timeout 600 python3 scripts/k2_stress.py 40 20000 2>&1 | tail -5 | cut -c1-400
