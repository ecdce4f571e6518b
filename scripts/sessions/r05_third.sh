#!/bin/bash
# round 5, third GPU call: does a side thread's hipMalloc of list-pool segments hold up the composition stage?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for side in 0 4 8; do
  echo "== 2 M reads, C3_SIDE_ALLOC=$side"
  C3_SIDE_ALLOC=$side C3_STAGE_CALLS=1 timeout 900 python3 scripts/c3_stage_probe.py 2000000 2>&1 | grep -v "^\[timing\]\|amdgpu.ids" | tail -12
done 2>&1 | tee gpurun_out/r05_side_alloc.txt
