#!/bin/bash
# rocprofv3 passes over the partitioned K2 accumulate (count, scan, part1, part2, slice)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_k2
mkdir -p "$OUT"
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -o k1 -- python3 scripts/k2_once.py > "$OUT/$name.log" 2>&1; echo "$name rc=$?"; }
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o k1 -- python3 scripts/k2_once.py > "$OUT/trace.log" 2>&1
run fetch FETCH_SIZE
run write WRITE_SIZE
run lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES
run wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE
run atom TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_STALL_sum
python3 scripts/pmc_summary.py "$OUT" "k15_" > gpurun_out/r02_k2_rocprof_summary.txt
cat gpurun_out/r02_k2_rocprof_summary.txt
