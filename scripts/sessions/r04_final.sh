#!/bin/bash
# end-of-round validation on the GPU box: the hard accuracy gate, the whole GPU suite, the K2/K3 profile, bench.py
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
T="tests/test_gpu_sim8.py::test_c1_hard_strains_vs_reference"
( time timeout 1200 python -m pytest "$T" -x -q -s > gpurun_out/r04_hard.log 2>&1 ) 2>&1 | grep real
grep -E "C1-hard e2e|passed|failed|assert" gpurun_out/r04_hard.log | cut -c1-200
( time timeout 3000 python -m pytest tests -m gpu -q --deselect "$T" > gpurun_out/r04_gpu_tests.log 2>&1 ) 2>&1 | grep real
tail -6 gpurun_out/r04_gpu_tests.log | cut -c1-200
bash scripts/prof_k2k3.sh > gpurun_out/r04_prof.log 2>&1; grep -E "rc=|Stop" gpurun_out/r04_prof.log | tr '\n' ' '; echo
grep -E "^wl_|kernel " gpurun_out/r04_k2k3_rocprof_summary.txt | head -20 | cut -c1-200
bash scripts/sessions/r04_bench.sh 2>&1 | grep -v test_gpu_multi
