#!/bin/bash
# round 2, first GPU call: the new reference-pinned tests + PMC passes of the k=4 LDS kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_vae_native.py -x -q -m gpu > gpurun_out/r02_vae_tests.log 2>&1
echo "vae tests rc=$?"
timeout 900 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu -k "contigs or resume or runner" > gpurun_out/r02_pipe_tests.log 2>&1
echo "pipeline tests rc=$?"
bash scripts/prof_k1.sh r02_k4lds --k 4
python3 scripts/pmc_summary.py gpurun_out/prof_r02_k4lds k1_ > gpurun_out/r02_k1_k4_lds_rocprof_summary.txt
tail -5 gpurun_out/r02_vae_tests.log gpurun_out/r02_pipe_tests.log
