#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
nproc; grep -c processor /proc/cpuinfo; numactl -H 2>/dev/null | head -3; cat /sys/fs/cgroup/cpu.max 2>/dev/null
( time timeout 1500 python3 scripts/c3_full.py > gpurun_out/r04_c3_full_b.json 2> gpurun_out/r04_c3_full_b.err ) 2>&1 | grep real
python3 -c "
import json
d=json.load(open('gpurun_out/r04_c3_full_b.json'))
print({k:(v.get('s') if isinstance(v,dict) else v) for k,v in d.items() if (isinstance(v,dict) and 's' in v) or k=='profiles_total_s'})"
