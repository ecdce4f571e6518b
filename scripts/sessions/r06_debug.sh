#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== one rank, 20000 reads"
timeout 600 python3 bench.py --steps 2 --warmup 1 --clock-ramp-ms 0 --reads 20000 --c4-reads 20000 --no-cpu-baseline --no-traffic --no-extra 2>&1 | tail -1 | cut -c1-1500
echo "== eight ranks"
LRB_BENCH_BACKEND=gloo timeout 1200 python3 bench.py --gpus 8 --steps 2 --warmup 1 --clock-ramp-ms 0 --reads 20000 --c4-reads 20000 --no-cpu-baseline --no-traffic --no-extra 2>&1 | tail -3 | cut -c1-3000
echo "== four ranks"
LRB_BENCH_BACKEND=gloo timeout 1200 python3 bench.py --gpus 4 --steps 2 --warmup 1 --clock-ramp-ms 0 --reads 20000 --c4-reads 20000 --no-cpu-baseline --no-traffic --no-extra 2>&1 | tail -1 | cut -c1-2000
timeout 900 python3 -m pytest tests/test_gpu_hdbscan.py -q -x -k "degenerate or identical" 2>&1 | tail -3
timeout 900 python3 -m pytest tests/test_gpu_vae_native.py -q -x 2>&1 | tail -2
