#!/bin/bash
# round 4, first GPU call: parity of the window-list route (part / order / tally / sweep of lrb_lists.hip), then
# kernel times of K2 + K3 at 400 k x 10 kb under rocprofv3, both occupancy variants of the order kernel
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "slice_lists or k3_sweep or from_slice" 2>&1 | tail -25 > gpurun_out/r04_first_tests.log
cat gpurun_out/r04_first_tests.log
for occ in 0 1; do
  OUT=gpurun_out/r04_trace_occ$occ
  rm -rf "$OUT"
  LRB_WL_ORDER_OCC=$occ timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o k -- python3 scripts/k2k3_once.py 400000 > "$OUT.log" 2>&1
  echo "occ=$occ rc=$?"; tail -3 "$OUT.log"
  python3 scripts/kstats.py "$OUT" wl_ | tee gpurun_out/r04_kstats_occ$occ.txt
done
