#!/bin/bash
# builds and runs scripts/k4s2_probe.hip on the GPU box: gpurun -- 'bash scripts/k4s2_probe.sh'
set -e
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ilrbinner_amd/csrc scripts/k4s2_probe.hip -Llrbinner_amd -llrb_hip \
      -Wl,-rpath,$PWD/lrbinner_amd -o gpurun_out/k4s2_probe
gpurun_out/k4s2_probe
