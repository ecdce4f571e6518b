#!/bin/bash
# round 5: the composition stage by parser-thread count and upload form on a box held to 16 CPUs (cgroup cpu.max)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for pack in 0 1; do for thr in 8 12 16 24 32; do
  echo "== LRB_HOST_PACK=$pack threads=$thr: $(LRB_HOST_PACK=$pack C3_THREADS=$thr C3_STAGE_CALLS=1 timeout 600 python3 scripts/c3_stage_probe.py 2000000 2>&1 | grep -A1 "^run_kmers" | tr '\n' ' ' | cut -c1-330)"
done; done 2>&1 | tee gpurun_out/r05_threads.txt
