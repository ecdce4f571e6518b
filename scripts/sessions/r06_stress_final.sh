#!/bin/bash
# round 6: with the count / part check in place -- lists tallied twice + atomic control, then K1 / K3, eight processes on one GPU
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "## k2_stress2.py: 8 processes x 150 passes (lists made once per pass, tallied twice; one-atomic-a-window control)"
for i in 0 1 2 3 4 5 6 7; do STRESS_TAG=$i timeout 1500 python3 scripts/k2_stress2.py 150 20000 > gpurun_out/sf_$i.log 2>&1 & done
wait
cat gpurun_out/sf_*.log | grep -v amdgpu.ids | cut -c1-220
echo "## k13_stress.py: 8 processes x 100 passes (K1 k = 3 / 4 / 5, K3 by both routes)"
for i in 0 1 2 3 4 5 6 7; do STRESS_TAG=$i timeout 1500 python3 scripts/k13_stress.py 100 20000 > gpurun_out/sf_$i.log 2>&1 & done
wait
cat gpurun_out/sf_*.log | grep -v amdgpu.ids | cut -c1-220
} | tee gpurun_out/r06_stress_final.txt
rm -f gpurun_out/sf_*.log
