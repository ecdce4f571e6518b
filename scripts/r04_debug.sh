#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2; do
  for v in old o2 o4 o8; do
    cp ab/liblrb_$v.so lrbinner_amd/liblrb_hip.so
    CFGS="$v$rep:" bash scripts/r04_time.sh 2>&1 | grep -E "order_kernel_occ1" | cut -c1-110 | tr '\n' ' '; echo $v
  done
done
