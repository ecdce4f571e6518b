#!/bin/bash
# rocprofv3 --kernel-trace --stats of the fused VAE training step (K7) at the batch sizes of the reference's schedule
# and both network shapes (k = 3: 42-128-128-4, k = 4: 168-128-128-8): kernel count and per-kernel time.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_vae
rm -rf "$OUT"; mkdir -p "$OUT"
summ() { python3 - "$1" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "vae_" in r["Name"]]
print("%-72s %7s %10s %9s %9s" % ("kernel", "calls", "avg_ns", "min_ns", "max_ns"))
tot = 0.0
for r in rows:
    print("%-72s %7s %10.0f %9s %9s" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"]))
    tot += float(r["AverageNs"]) * int(r["Calls"])
steps = max(int(r["Calls"]) for r in rows if "adam" in r["Name"])
print("kernel time per step: %.1f us over %d steps (%d kernel launches per step)" % (tot / steps / 1e3, steps, round(sum(int(r["Calls"]) for r in rows) / steps)))
PY
}
{
for shape in "10 32 4" "32 136 8"; do
  for bs in 1024 2048 4096 8192; do
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/t_${bs}_${shape// /_}" -o v -- python3 scripts/vae_native_trace.py $bs $shape > "$OUT/log_${bs}.txt" 2>&1
    echo "# scripts/vae_native_trace.py $bs $shape  (cov prof latent; hidden 128,128; $((200000 / bs)) steps of $bs rows)"
    summ "$OUT/t_${bs}_${shape// /_}/v_kernel_stats.csv"
    echo
  done
done
} > gpurun_out/r03_vae_rocprof.txt
cat gpurun_out/r03_vae_rocprof.txt
