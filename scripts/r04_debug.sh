#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 scripts/ubench_scatter_write.hip -o gpurun_out/ubench_scatter_write 2>&1 | tail -3
timeout 300 gpurun_out/ubench_scatter_write | tee gpurun_out/r04_ubench_scatter_write.txt
rm -f gpurun_out/ubench_scatter_write
