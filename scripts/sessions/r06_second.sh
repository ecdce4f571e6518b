#!/bin/bash
# round 6, session 2: the GPU suite over the round's new code, the dW kernel A/B, then the accuracy comparison under EQUAL search seeds
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 3000 python3 -m pytest tests -q -m gpu -rf 2>&1 | tail -40 | tee gpurun_out/r06_gpu_suite_a.txt
timeout 600 python3 scripts/r06_vae_dw_ab.py 2>&1 | tail -8
export R06_FIRST_STEP=400
R06_TAG=_s1001 R06_SEARCH_SEEDS=1001,1002,1003 R06_LATENTS=_ship/ref_latents_hard timeout 900 python3 scripts/r06_accuracy_runs.py c1hard 0 3 > gpurun_out/r06_ref_recluster_b.log 2>&1
tail -2 gpurun_out/r06_ref_recluster_b.log | cut -c1-400
R06_TAG=_s1to8 R06_SEARCH_SEEDS=1,2,3,4,5,6,7,8 timeout 1800 python3 scripts/r06_accuracy_runs.py c1hard 40 8 > gpurun_out/r06_c1hard_runs_b.log 2>&1
tail -1 gpurun_out/r06_c1hard_runs_b.log | cut -c1-400
R06_VAE=torch R06_TAG=_s1to8 R06_SEARCH_SEEDS=1,2,3,4,5,6,7,8 timeout 2700 python3 scripts/r06_accuracy_runs.py c1hard 40 8 > gpurun_out/r06_c1hard_runs_torch_b.log 2>&1
tail -1 gpurun_out/r06_c1hard_runs_torch_b.log | cut -c1-400
