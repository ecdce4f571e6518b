#!/bin/bash
# kernel times of K2 + K3 (window lists) at 400 k x 10 kb under rocprofv3, for the experiment switches given as
# "name:ENV=val,ENV=val" words in CFGS
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for cfg in ${CFGS:-base:}; do
  name=${cfg%%:*}; envs=${cfg#*:}
  OUT=gpurun_out/r04_trace_$name
  rm -rf "$OUT"
  ( IFS=,; for e in $envs; do [ -n "$e" ] && export "$e"; done
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o k -- python3 scripts/k2k3_once.py ${K2K3_N:-400000} > "$OUT.log" 2>&1
    echo "$name rc=$?" )
  grep -i "error\|assert" "$OUT.log" | head -3
  python3 scripts/kstats.py "$OUT" wl_ | grep -v "gscan\|gbase" | tee gpurun_out/r04_kstats_$name.txt
done
