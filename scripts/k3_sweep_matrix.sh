#!/bin/bash
# K3 sweep: kernel times by reads per group and workgroups per CU (rocprofv3 kernel trace).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_k3_sweep
mkdir -p "$OUT"
# SWEEP_CFGS: space-separated reads:workgroups-per-CU pairs
for cfg in ${SWEEP_CFGS:-391:2 261:2 196:2 157:2 782:2 391:1 1563:1}; do
    set -- ${cfg/:/ }
    export LRB_K3_SWEEP_READS=$1 LRB_K3_SWEEP_PER_CU=$2
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/m_$1_$2" -o k1 -- python3 scripts/k3_sweep_once.py ${SWEEP_N:-400000} > "$OUT/m_$1_$2.log" 2>&1
    python3 - "$OUT/m_$1_$2" "$1" "$2" <<'PY'
import csv, glob, sys
t = {}
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "cov_join" in row["Name"]:
            t[row["Name"].split("(")[0]] = float(row["AverageNs"]) / 1e6
print(f"reads/group {sys.argv[2]:>5s}  workgroups/CU {sys.argv[3]}:  part {t.get('cov_join_part_kernel', 0):7.3f} ms  sweep {t.get('cov_join_sweep_kernel', 0):7.3f} ms")
PY
done
