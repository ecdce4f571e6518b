#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2; do
  for v in old nocarry new; do
    cp ab/liblrb_$v.so lrbinner_amd/liblrb_hip.so
    CFGS="$v$rep:" bash scripts/r04_time.sh 2>&1 | grep -E "rc=|part" | cut -c1-110
  done
done
# HBM bytes written by the part kernel, new build
cp ab/liblrb_new.so lrbinner_amd/liblrb_hip.so
for c in WRITE_SIZE "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_w -o k -- python3 scripts/k2k3_once.py 400000 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/pmc_w/*counter_collection.csv'):
    acc={}
    for r in csv.DictReader(open(f)):
        if 'wl_part' in r['Kernel_Name']:
            acc.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
    print({k:sum(v)/len(v) for k,v in acc.items()})
PY
rm -rf gpurun_out/pmc_w
done
