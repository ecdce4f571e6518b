#!/bin/bash
# round 6, session 1: accuracy evidence with a continuous statistic (whole runs + per-latent re-clustering)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
R06_LATENTS=_ship/ref_latents_hard timeout 900 python3 scripts/r06_accuracy_runs.py c1hard 0 8 > gpurun_out/r06_ref_recluster.log 2>&1
tail -3 gpurun_out/r06_ref_recluster.log | cut -c1-300
timeout 2400 python3 scripts/r06_accuracy_runs.py c1hard 100 3 > gpurun_out/r06_c1hard_runs.log 2>&1
tail -2 gpurun_out/r06_c1hard_runs.log | cut -c1-300
timeout 900 python3 scripts/r06_accuracy_runs.py c1 40 3 > gpurun_out/r06_c1_runs.log 2>&1
tail -2 gpurun_out/r06_c1_runs.log | cut -c1-300
