#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python3 scripts/k4_seed_hist_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_k4_probe.txt
rm -rf /tmp/k4trace; export K4_ONLY_C1=1; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k4trace -o k -- python3 scripts/k4_seed_hist_probe.py > /dev/null 2>&1
python3 scripts/kstats.py /tmp/k4trace seed_ | tee gpurun_out/r05_k4_kstats.txt
