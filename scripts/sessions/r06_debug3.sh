#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for i in 1 2; do timeout 1200 python3 -m pytest tests/test_gpu_sim8.py -x -q -k "sim8_end_to_end or sim8_reference_latents" 2>&1 | tail -40 | cut -c1-300; done
