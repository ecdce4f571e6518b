#!/bin/bash
# same-box A/B of the two K4 forms: every pair's increment made (out of range -> a row nobody reads) against increments under the
# range test's exec mask; on one blob (most pairs within 0.3) and on eight (an eighth of them)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp K4_ONLY_C1=1
mkdir -p gpurun_out
cp lrbinner_amd/liblrb_hip.so /tmp/keep.so
for rep in 1 2; do
  for v in trash masked; do
    cp ab/liblrb_$v.so lrbinner_amd/liblrb_hip.so
    for blobs in "" 1; do
      echo "$v rep $rep blobs=${blobs:-0}: $(K4_BLOBS=$blobs timeout 300 python3 scripts/k4_seed_hist_probe.py 2>&1 | grep dims)"
    done
  done
done | tee gpurun_out/r05_k4_ab.txt
cp /tmp/keep.so lrbinner_amd/liblrb_hip.so
