#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp R06_FIRST_STEP=400
mkdir -p gpurun_out
R06_TAG=_s1to8_more R06_SEARCH_SEEDS=1,2,3,4,5,6,7,8 R06_LATENTS=_ship/ref_latents_c1_new timeout 900 python3 scripts/r06_accuracy_runs.py c1 0 8 > gpurun_out/r06_c1_ref_more.log 2>&1
tail -2 gpurun_out/r06_c1_ref_more.log | cut -c1-300
