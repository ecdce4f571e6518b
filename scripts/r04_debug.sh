#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
export LRB_WL_PART=2
for v in old prio old prio; do
  cp ab/liblrb_$v.so lrbinner_amd/liblrb_hip.so
  CFGS="$v:" bash scripts/r04_time.sh 2>&1 | grep -E "part" | cut -c1-110 | tr '\n' ' '; echo $v
done
