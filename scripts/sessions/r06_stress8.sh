#!/bin/bash
# the eight-rank bench command again and again on one GPU: does the result check ever fail, and on what?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp LRB_BENCH_BACKEND=gloo
mkdir -p gpurun_out
for i in $(seq 1 12); do
  timeout 600 python3 bench.py --gpus 8 --steps 2 --warmup 1 --clock-ramp-ms 0 --reads 20000 --c4-reads 20000 --no-cpu-baseline --no-traffic --no-extra 2>/dev/null | tail -1 | python3 -c "
import sys, json
l = sys.stdin.read().strip()
try:
    d = json.loads(l); c = d['roofline']['c4_rank']
    print('run $i', 'ERROR ' + str(c['error']) if 'error' in c else 'ok %.0f reads/s' % c['default_reads_per_s'])
except Exception as e:
    print('run $i no line', l[:200])
"
done | tee gpurun_out/r06_stress8.txt
