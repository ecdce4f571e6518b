#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_bins.py tests/test_gpu_c5.py -x -q 2>&1 | tail -3
for n in 2000000 5000000; do
  echo "== $n reads, defaults"
  C3_STAGE_CALLS=1 timeout 1200 python3 scripts/c3_stage_probe.py $n 2>&1 | grep -v "^\[timing\]\|amdgpu.ids" | tail -9
done 2>&1 | tee gpurun_out/r05_c3_stage_calls_final.txt
timeout 2400 bash scripts/prof_r05.sh 2>&1 | grep "rc=" | tr '\n' ' '
