#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out

( time timeout 1500 python bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err ) 2>&1 | tail -4
tail -3 gpurun_out/r04_bench.err
python3 - <<'PY'
import json
l=[x for x in open('gpurun_out/r04_bench.json') if x.startswith('{')]
d=json.loads(l[-1])
print(json.dumps({k:d[k] for k in ('value','ms_per_step')}), json.dumps(d['roofline']['frac']))
rs=d.get('roofline_stages',{})
for k,v in rs.items():
    if isinstance(v,dict): print(k, {kk:(round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ('kernel_ms','frac','achieved','frac_hbm','traffic','error')})
    else: print(k, v)
print(json.dumps(d.get('cpu_baseline'), indent=1)[:3000])
c4=d.get('c4_phases',{}); print({k:c4.get(k) for k in ('reads_per_s','phases_ms_max_over_ranks','error')})
PY
