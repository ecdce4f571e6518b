#!/bin/bash
# the line `bench.py --gpus 8` prints (gloo ranks on one GPU: the numbers mean nothing, the format is what a node will print)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp LRB_BENCH_BACKEND=gloo
mkdir -p gpurun_out
LRB_BENCH_DETAIL=gpurun_out/r06_bench_gpus8_gloo_detail.json LRB_COLLECTIVE_TIMEOUT_S=120 timeout 600 python3 bench.py --gpus 8 --steps 20 --warmup 5 --reads 50000 --c4-reads 40000 --no-traffic --no-extra > gpurun_out/r06_bench_gpus8_gloo.json 2> gpurun_out/r06_bench_gpus8_gloo.err; echo "rc=$?"
python3 -c "
import json; d=json.loads(open('gpurun_out/r06_bench_gpus8_gloo.json').read().strip().splitlines()[-1]); print(len(json.dumps(d)), d['n_gpus'], d['roofline']['c4_rank'])"
tail -3 gpurun_out/r06_bench_gpus8_gloo.err | cut -c1-300
