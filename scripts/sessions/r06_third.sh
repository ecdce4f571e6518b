#!/bin/bash
# round 6, session 3: the evidence files -- rocprofv3 passes behind every fraction of the line, the default bench line + its
# detail file, the eight-rank tail probe, the reference's latents under their own search seeds, C3 / C5 at full size
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_hdbscan.py -q -k "degenerate or identical" 2>&1 | tail -25 | cut -c1-250
for i in 1 2 3; do timeout 1500 python3 -m pytest tests/test_gpu_multi.py -q -k "bench_gpus_8" 2>&1 | grep -E "passed|failed|AssertionError|result check" | cut -c1-900; done
timeout 2400 bash scripts/prof_r06.sh > gpurun_out/prof_r06.log 2>&1; tail -3 gpurun_out/prof_r06.log
for f in gpurun_out/r06_k1_k4_warm_rocprof_summary.txt gpurun_out/r06_k1_k5_warm_rocprof_summary.txt gpurun_out/r06_k1_lane_rocprof_summary.txt; do grep -E "avg_ns" $f | cut -c1-150; done
LRB_BENCH_DETAIL=gpurun_out/r06_bench_detail.json timeout 1500 python3 bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err; echo "bench rc=$?"; tail -2 gpurun_out/r06_bench.err
python3 - <<'P'
import json
d=json.loads(open('gpurun_out/r06_bench.json').read().strip().splitlines()[-1])
print(len(json.dumps(d)), d['value'], d['roofline']['frac'])
for k,v in d['roofline']['stages'].items(): print(k, v)
print(d['roofline'].get('c4_rank'))
P
timeout 1500 python3 scripts/dist_tail_probe.py 800000 1,8 2>&1 | tail -3 | cut -c1-600
R06_OWN_SEED=1 R06_TAG=_own R06_LATENTS=_ship/ref_latents_hard timeout 600 python3 scripts/r06_accuracy_runs.py c1hard 0 1 > gpurun_out/r06_ref_own.log 2>&1; tail -3 gpurun_out/r06_ref_own.log | cut -c1-200
timeout 900 python3 scripts/c3_full.py > gpurun_out/r06_c3_full.json 2> gpurun_out/r06_c3_full.err; echo "c3 rc=$?"; tail -c 600 gpurun_out/r06_c3_full.json
timeout 900 python3 scripts/c5_full.py > gpurun_out/r06_c5_full.json 2> gpurun_out/r06_c5_full.err; echo "c5 rc=$?"; tail -c 400 gpurun_out/r06_c5_full.json
