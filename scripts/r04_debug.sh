#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for d in 0 16 1; do
OUT=gpurun_out/r04_trace_dbg$d
rm -rf $OUT
LRB_WL_DBG=$d timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o k -- python3 scripts/k2k3_time_only.py 400000 > "$OUT.log" 2>&1
echo "dbg=$d $(python3 scripts/kstats.py "$OUT" wl_part)"
done
