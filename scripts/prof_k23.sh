#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_k23
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o k1 -- python3 scripts/k23_once.py > "$OUT/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o k1 -- python3 scripts/k23_once.py > "$OUT/f.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o k1 -- python3 scripts/k23_once.py > "$OUT/w.log" 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum TCC_EA0_RDREQ_sum --output-format csv -d "$OUT/pmc_tcc" -o k1 -- python3 scripts/k23_once.py > "$OUT/t.log" 2>&1
python3 scripts/pmc_summary.py "$OUT" "k" | grep -v "at::native" 
