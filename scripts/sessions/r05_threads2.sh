#!/bin/bash
# round 5: the composition stage after the multi-threaded profile writer, by upload form and thread count; then all three
# stages at 2 M and 5 M reads with the defaults
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_bins.py -x -q 2>&1 | tail -3
for pack in 0 1; do for thr in 8 32; do
  echo "== LRB_HOST_PACK=$pack threads=$thr: $(LRB_HOST_PACK=$pack C3_THREADS=$thr C3_STAGE_CALLS=1 timeout 600 python3 scripts/c3_stage_probe.py 2000000 2>&1 | grep -A1 "^run_kmers" | tr '\n' ' ' | cut -c1-330)"
done; done 2>&1 | tee gpurun_out/r05_threads2.txt
for n in 2000000 5000000; do
  echo "== $n reads, defaults"
  C3_STAGE_CALLS=1 timeout 1200 python3 scripts/c3_stage_probe.py $n 2>&1 | grep -v "^\[timing\]\|amdgpu.ids" | tail -9
done 2>&1 | tee gpurun_out/r05_c3_stage_calls_final.txt
