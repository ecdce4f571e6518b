#!/bin/bash
# round 5: the determinism test, the rocprofv3 evidence for every frac, the merge rates on the hard set, the multi-rank tail
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_vae_native.py -x -q 2>&1 | tail -5
timeout 2400 bash scripts/prof_r05.sh 2>&1 | tail -60
timeout 1500 python3 scripts/dist_tail_probe.py 2000000 2>&1 | grep -v amdgpu.ids | tail -6
timeout 2400 python3 scripts/c1_hard_rates.py 60 2>&1 | grep -v amdgpu.ids | tail -12
