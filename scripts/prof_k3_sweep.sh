#!/bin/bash
# rocprofv3 kernel times of the K3 sweep (part + sweep kernels) by reads per group.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_k3_sweep
mkdir -p "$OUT"
for R in ${SWEEP_RS:-0 256 512 1024}; do
    if [ "$R" != 0 ]; then export LRB_K3_SWEEP_READS=$R; else unset LRB_K3_SWEEP_READS; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$R" -o k1 -- python3 scripts/k3_sweep_once.py ${SWEEP_N:-400000} > "$OUT/trace_$R.log" 2>&1
    echo "== reads per group: $R (0 = default) =="
    python3 - "$OUT/trace_$R" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "cov_" in row["Name"]:
            print(f'{row["Name"][:40]:40s} calls={row["Calls"]:>3s} avg_ms={float(row["AverageNs"])/1e6:8.3f}')
PY
done
