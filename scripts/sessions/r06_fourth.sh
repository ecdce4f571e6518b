#!/bin/bash
# round 6, session 4: where did the table / coverage phases of a C4 rank lose 20 ms (the one-kernel concat?)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python3 -m pytest tests/test_gpu_hdbscan.py -q -k "degenerate or identical" 2>&1 | tail -12 | cut -c1-250
rm -rf gpurun_out/prof_c4gap; mkdir -p gpurun_out/prof_c4gap
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c4gap -o k1 -- python3 scripts/c4_gap_probe.py > gpurun_out/prof_c4gap/log.txt 2>&1
tail -3 gpurun_out/prof_c4gap/log.txt
python3 - <<'P'
import csv, glob
f = glob.glob('gpurun_out/prof_c4gap/**/k1_kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(f"{r['Name'][:60]:60s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} total_ms={float(r['TotalDurationNs'])/1e6:9.2f}")
P
for b in 32 64 256; do echo "LRB_CONCAT_BLOCKS=$b"; LRB_CONCAT_BLOCKS=$b timeout 300 python3 scripts/c4_gap_probe.py 2>&1 | tail -2; done
echo "LRB_RESIDENT_LISTS=0"; LRB_RESIDENT_LISTS=0 timeout 300 python3 scripts/c4_gap_probe.py 2>&1 | tail -2
