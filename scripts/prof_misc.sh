#!/bin/bash
# rocprofv3 --kernel-trace --stats of the partitioned K2 accumulate, the fused VAE step and the
# HDBSCAN kernels (run on the GPU box through gpurun); summaries -> gpurun_out/prof_misc/*.txt
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_misc
mkdir -p "$OUT"
summ() { python3 - "$1" "$2" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if any(t in r["Name"] for t in sys.argv[2].split(","))]
print("%-64s %7s %12s %10s %10s" % ("kernel", "calls", "avg_ns", "min_ns", "max_ns"))
for r in keep:
    print("%-64s %7s %12.0f %10s %10s" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"]))
PY
}
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/k2" -o k2 -- python3 scripts/k2_probe.py 400000 > "$OUT/k2.log" 2>&1
{ echo "# scripts/k2_probe.py 400000 (400 k reads x 10 kb = 4.0e9 15-mers per accumulate)"; grep "n=" "$OUT/k2.log"; summ "$OUT/k2/k2_kernel_stats.csv" "k15_"; } > "$OUT/r01_k2_partitioned_rocprof.txt"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/vae" -o v -- python3 scripts/vae_native_trace.py 1024 > "$OUT/vae.log" 2>&1
{ echo "# scripts/vae_native_trace.py 1024 (195 steps of 1024 rows, network 42-128-128-4-128-128-42)"; summ "$OUT/vae/v_kernel_stats.csv" "vae_"; } > "$OUT/r01_vae_rocprof.txt"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/hdb" -o h -- python3 scripts/hdb_probe.py 500000 > "$OUT/hdb.log" 2>&1
{ echo "# scripts/hdb_probe.py 500000"; grep "n=" "$OUT/hdb.log"; summ "$OUT/hdb/h_kernel_stats.csv" "hdb_"; } > "$OUT/r01_hdbscan_rocprof.txt"
cat "$OUT"/r01_*.txt
