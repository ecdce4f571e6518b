#!/bin/bash
# round 5, second GPU call: the three profile stages at C3's read shape, wall time per library entry point, with the slice
# lists of the table stage kept for the coverage stage (LRB_KEEP_LISTS=1) and not; then at C3's full size
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for keep in 0 1; do
  echo "== 2 M reads, LRB_KEEP_LISTS=$keep"
  LRB_KEEP_LISTS=$keep C3_STAGE_CALLS=1 timeout 900 python3 scripts/c3_stage_probe.py 2000000 2>&1 | grep -v "^\[timing\]\|amdgpu.ids" | tail -12
done 2>&1 | tee gpurun_out/r05_c3_stage_calls_keep.txt
for keep in 0 1; do
  echo "== 5 M reads, LRB_KEEP_LISTS=$keep"
  LRB_KEEP_LISTS=$keep C3_STAGE_CALLS=1 timeout 1200 python3 scripts/c3_stage_probe.py 5000000 2>&1 | grep -v "^\[timing\]\|amdgpu.ids" | tail -12
done 2>&1 | tee gpurun_out/r05_c3_stage_calls_5m.txt
