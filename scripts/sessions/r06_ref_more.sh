#!/bin/bash
# round 6: the reference latents made since session 2 (runs 26 ...) under the same search seeds 1-8, with the first-step
# statistic, and each under its own run's seed
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
R06_FIRST_STEP=400 R06_TAG=_s1to8_more R06_SEARCH_SEEDS=1,2,3,4,5,6,7,8 R06_LATENTS=_ship/ref_latents_hard_new timeout 1500 python3 scripts/r06_accuracy_runs.py c1hard 0 8 > gpurun_out/r06_ref_more.log 2>&1
tail -2 gpurun_out/r06_ref_more.log | cut -c1-300
R06_OWN_SEED=1 R06_TAG=_own_more R06_LATENTS=_ship/ref_latents_hard_new timeout 600 python3 scripts/r06_accuracy_runs.py c1hard 0 1 > gpurun_out/r06_ref_own_more.log 2>&1
tail -2 gpurun_out/r06_ref_own_more.log | cut -c1-200
