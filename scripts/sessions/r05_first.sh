#!/bin/bash
# round 5, first GPU call: smoke, what a big hipMalloc costs and whom it holds up (scripts/ubench_malloc.hip), and the
# three profile stages at C3's read shape with the wall time per library entry point
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 300 scripts/bin/ubench_malloc 2>&1 | tee gpurun_out/r05_ubench_malloc.txt
C3_STAGE_CALLS=1 LRB_TIMING=1 timeout 900 python3 scripts/c3_stage_probe.py 2000000 2>&1 | grep -v "^\[timing\] cov_hist_many" | tail -40 | tee gpurun_out/r05_c3_stage_calls.txt
