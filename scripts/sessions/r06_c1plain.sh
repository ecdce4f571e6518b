#!/bin/bash
# round 6: the equal-search-seeds comparison on the plain C1 stand-in (BASELINE config C1's own size and flags)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp R06_FIRST_STEP=400
mkdir -p gpurun_out
R06_TAG=_s1to8 R06_SEARCH_SEEDS=1,2,3,4,5,6,7,8 R06_LATENTS=_ship/ref_latents_c1 timeout 900 python3 scripts/r06_accuracy_runs.py c1 0 8 > gpurun_out/r06_c1_ref_recluster.log 2>&1
tail -2 gpurun_out/r06_c1_ref_recluster.log | cut -c1-300
R06_TAG=_s1to8 R06_SEARCH_SEEDS=1,2,3,4,5,6,7,8 timeout 1800 python3 scripts/r06_accuracy_runs.py c1 40 8 > gpurun_out/r06_c1_runs_b.log 2>&1
tail -1 gpurun_out/r06_c1_runs_b.log | cut -c1-300
