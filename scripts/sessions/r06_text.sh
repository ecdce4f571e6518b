#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
LRB_BENCH_DETAIL=gpurun_out/r06_bench_detail.json timeout 1500 python3 bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err; echo "bench rc=$?"; tail -2 gpurun_out/r06_bench.err | cut -c1-300
python3 - <<'P'
import json
d=json.loads(open('gpurun_out/r06_bench.json').read().strip().splitlines()[-1])
print(len(json.dumps(d)), d['value'], d['roofline']['frac'])
print(d['roofline'].get('c4_rank'))
P
timeout 1200 python3 -m pytest tests/test_gpu_multi.py -q -k "bench_gpus" 2>&1 | tail -3
