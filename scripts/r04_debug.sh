#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "slice_lists or k3_sweep or from_slice or c4_rank or c3_full" > gpurun_out/r04_tests_full.log 2>&1
grep -v "^  File\|^Extension" gpurun_out/r04_tests_full.log | tail -4
bash scripts/prof_k2k3.sh > gpurun_out/r04_prof.log 2>&1; grep "kernel-trace" -A8 gpurun_out/r04_k2k3_rocprof_summary.txt | cut -c1-160
( time timeout 1500 python bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err ) 2>&1 | tail -4
python3 - <<'PY'
import json
l=[x for x in open('gpurun_out/r04_bench.json') if x.startswith('{')]
d=json.loads(l[-1])
print(json.dumps({k:d[k] for k in ('value','ms_per_step')}), d['roofline']['frac'])
rs=d.get('roofline_stages',{})
for k,v in rs.items():
    if isinstance(v,dict): print(k, {kk:(round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ('kernel_ms','frac','achieved','frac_hbm','traffic','error')})
c4=d.get('c4_phases',{}); print({k:c4.get(k) for k in ('reads_per_s','phases_ms_max_over_ranks','error')})
PY
