#!/bin/bash
# round 5: host-side packing -- the new parity test, the pipeline tests that now run through it, the parser pool alone
# (ASCII against packed batches by thread count), the three profile stages with call times, the tally16 micro-benchmark
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "host_packed or pack_matches or k2_half_route" 2>&1 | tail -4
timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_bins.py tests/test_gpu_multi.py -x -q 2>&1 | tail -4
timeout 300 scripts/bin/ubench_tally16 2>&1 | tee gpurun_out/r05_ubench_tally16.txt
timeout 900 python3 scripts/parser_probe.py 2000000 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_parser_probe.txt
for n in 2000000 5000000; do
  echo "== $n reads"
  C3_STAGE_CALLS=1 timeout 1200 python3 scripts/c3_stage_probe.py $n 2>&1 | grep -v "^\[timing\]\|amdgpu.ids" | tail -9
done 2>&1 | tee gpurun_out/r05_c3_stage_calls_packed.txt
