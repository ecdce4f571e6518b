#!/bin/bash
# round 6: the rocprofv3 pass set once more on the final kernels (the part kernel carries the count check now)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 bash scripts/prof_r06.sh > gpurun_out/prof_r06.log 2>&1; tail -2 gpurun_out/prof_r06.log
for f in gpurun_out/r06_k1_k4_warm_rocprof_summary.txt gpurun_out/r06_k1_k5_warm_rocprof_summary.txt gpurun_out/r06_k1_lane_rocprof_summary.txt; do grep -E "avg_ns" $f | cut -c1-150; done
grep -E "avg_ns" gpurun_out/r06_k2k3_rocprof_summary.txt | cut -c1-150
rm -rf gpurun_out/prof_r06_stages gpurun_out/prof_r06_k1 gpurun_out/prof_r06_k1k4 gpurun_out/prof_r06_k1k5
du -sh gpurun_out
