#!/bin/bash
# round 6: every barrier behind lgkmcnt(0) -- the suite, the line, the C3 stages, the stress (partitions repeated should be 0)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 3000 python3 -m pytest tests -q -m gpu -rf > gpurun_out/r06_gpu_suite_full.txt 2>&1; tail -6 gpurun_out/r06_gpu_suite_full.txt | head -3 | tee gpurun_out/r06_gpu_suite.txt
LRB_BENCH_DETAIL=gpurun_out/r06_bench_detail.json timeout 1500 python3 bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err; echo "bench rc=$?"
python3 - <<'P'
import json
d=json.loads(open('gpurun_out/r06_bench.json').read().strip().splitlines()[-1])
print(len(json.dumps(d)), d['value'], d['roofline']['frac'], {k:(v['frac'],v['kernel_ms']) for k,v in d['roofline']['stages'].items()})
print(d['roofline'].get('c4_rank'))
P
timeout 900 python3 scripts/c3_stage_probe.py 5000000 2>&1 | tail -4 | tee gpurun_out/r06_c3_stage_calls.txt
export STRESS_TEXT=0
{
for rep in 1 2; do
  for i in 0 1 2 3 4 5 6 7; do STRESS_TAG=$i timeout 1200 python3 scripts/k2_stress.py 80 20000 > gpurun_out/sb_$i.log 2>&1 & done
  wait
  cat gpurun_out/sb_*.log | grep -v amdgpu.ids | grep -E "passes|short" | cut -c1-200
done
} | tee gpurun_out/r06_k2_stress_barrier.txt
rm -rf gpurun_out/sb_*.log gpurun_out/sim8_latents
