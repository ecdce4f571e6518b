#!/bin/bash
# BASELINE C3 and C5 at full size through the runner / CLI paths (file -> file), for the round's record
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1500 python3 scripts/c3_full.py > gpurun_out/r04_c3_full.json 2> gpurun_out/r04_c3_full.err ) 2>&1 | grep real
tail -c 1500 gpurun_out/r04_c3_full.json; echo
( time timeout 1500 python3 scripts/c5_full.py > gpurun_out/r04_c5_full.json 2> gpurun_out/r04_c5_full.err ) 2>&1 | grep real
tail -c 1200 gpurun_out/r04_c5_full.json; echo
