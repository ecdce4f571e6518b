#!/bin/bash
# the whole GPU suite + smoke + the default bench line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 4500 python -m pytest tests -q -m gpu 2>&1 | tail -25 | tee gpurun_out/r05_gpu_suite.txt
timeout 1500 python bench.py > gpurun_out/r05_bench_b.json 2> gpurun_out/r05_bench_b.err; echo "bench rc=$?"; tail -3 gpurun_out/r05_bench_b.err
