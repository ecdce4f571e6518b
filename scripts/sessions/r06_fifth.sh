#!/bin/bash
# round 6, session 5: the workspaces reserved for the largest group -- C4 rank again, the three stages at C3 size, the line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python3 scripts/c4_gap_probe.py 2>&1 | tail -3
timeout 600 python3 -m pytest tests/test_gpu_pipeline.py -q -k "last_groups or keeps_its_slice or sweep_over_many or sharded" 2>&1 | tail -3
timeout 600 python3 -m pytest tests/test_gpu_parity.py -q -k "many_resident or c4 or slice_lists or lists" 2>&1 | tail -3
timeout 900 python3 scripts/c3_stage_probe.py 5000000 2>&1 | tail -8 | tee gpurun_out/r06_c3_stage_calls.txt
LRB_BENCH_DETAIL=gpurun_out/r06_bench_detail.json timeout 1500 python3 bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err; echo "bench rc=$?"
python3 - <<'P'
import json
d=json.loads(open('gpurun_out/r06_bench.json').read().strip().splitlines()[-1])
print(len(json.dumps(d)), d['value'], d['roofline']['frac'])
for k in ('k1_k4','k1_k5','k2','k3_default','k3_kept_lists'): print(k, d['roofline']['stages'][k])
print(d['roofline'].get('c4_rank'))
P
