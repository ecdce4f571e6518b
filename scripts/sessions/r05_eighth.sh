#!/bin/bash
# after the part kernel's ring form became the only one: full GPU suite, K2 / K3 profile, driver line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | head -5 | tee gpurun_out/r05_gpu_suite.txt
bash scripts/prof_k2k3.sh > gpurun_out/prof_k2k3.log 2>&1; cp gpurun_out/r04_k2k3_rocprof_summary.txt gpurun_out/r05_k2k3_rocprof_summary.txt
grep -E "avg_ns" gpurun_out/r05_k2k3_rocprof_summary.txt | cut -c1-130
timeout 1500 python3 bench.py > gpurun_out/r05_bench_c.json 2> gpurun_out/r05_bench_c.err; echo "bench rc=$?"
python3 - <<'P'
import json
d=json.loads(open('gpurun_out/r05_bench_c.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'])
for k,v in d['roofline_stages'].items():
    if 'kernel_ms' in v: print(k, v.get('kernel_ms'), v.get('frac'), v.get('traffic'))
c=d['c4_phases']
for r,v in c['routes'].items(): print(r, v['phases_ms_max_over_ranks'], v['reads_per_s'])
P
