#!/bin/bash
# same-box A/B of two builds of liblrb_hip.so (ab/liblrb_old.so, ab/liblrb_new.so), alternating, K2/K3 kernel times
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2; do
  for v in old new; do
    cp ab/liblrb_$v.so lrbinner_amd/liblrb_hip.so
    CFGS="$v$rep:" bash scripts/sessions/r04_time.sh 2>&1 | grep -E "rc=|${PAT:-part|order_kernel_occ1|count|tally|sweep}" | cut -c1-110
  done
done
