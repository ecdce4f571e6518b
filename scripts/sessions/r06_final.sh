#!/bin/bash
# round 6, final evidence on the final tree: the whole GPU suite, the C4 rank and a whole C1 run under rocprofv3, the line,
# the equal-search-seeds comparison on the plain C1 stand-in
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 3000 python3 -m pytest tests -q -m gpu -rf > gpurun_out/r06_gpu_suite_full.txt 2>&1; tail -15 gpurun_out/r06_gpu_suite_full.txt | tee gpurun_out/r06_gpu_suite.txt; grep -n -i "error\|LrbError\|disagree\|out of memory" gpurun_out/r06_gpu_suite_full.txt | head -20
rm -rf gpurun_out/prof_c4gap; mkdir -p gpurun_out/prof_c4gap
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c4gap -o k1 -- python3 scripts/c4_gap_probe.py > gpurun_out/prof_c4gap/log.txt 2>&1
python3 - <<'P' | tee gpurun_out/r06_c4_rank_kernel_stats.txt
import csv, glob
print("# scripts/c4_gap_probe.py (a C4-shaped rank: 2.5 M reads of 10 kb in 373 resident batches, table phase + coverage phase through lrbinner_amd.dist.HipCompute, three passes) under rocprofv3 --kernel-trace --stats")
print(open('gpurun_out/prof_c4gap/log.txt').read().strip().splitlines()[-3:])
f = glob.glob('gpurun_out/prof_c4gap/**/k1_kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r['TotalDurationNs']) > 1e5:
        print(f"{r['Name'].split('(')[0][:60]:60s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} total_ms={float(r['TotalDurationNs'])/1e6:9.2f}")
P
sed -e 's/r05_pipeline_kernel_stats/r06_pipeline_kernel_stats/g' scripts/sessions/r05_pipeline_prof.sh > /tmp/pipe_r06.sh; bash /tmp/pipe_r06.sh 2>&1 | tail -22 | cut -c1-150
LRB_BENCH_DETAIL=gpurun_out/r06_bench_detail.json timeout 1500 python3 bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err; echo "bench rc=$?"
python3 - <<'P'
import json
d=json.loads(open('gpurun_out/r06_bench.json').read().strip().splitlines()[-1])
print(len(json.dumps(d)), d['value'], d['roofline']['frac'], {k:v['frac'] for k,v in d['roofline']['stages'].items() if k.startswith('k')})
print(d['roofline'].get('c4_rank'))
P
rm -rf gpurun_out/prof_c4gap gpurun_out/pipe_trace gpurun_out/pipe.log
bash scripts/sessions/r06_c1plain.sh
rm -rf gpurun_out/sim8_latents; du -sh gpurun_out
