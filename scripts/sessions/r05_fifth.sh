#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python3 scripts/k4_seed_hist_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_k4_probe.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py -x -q -k "k4 or seed or cluster" 2>&1 | tail -5
