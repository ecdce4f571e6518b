#!/bin/bash
# rocprofv3 passes over the K3 sweep (cov_join_part_kernel + cov_join_sweep_kernel) at 400 k x 10 kb:
# kernel times, L2 hits / misses, HBM traffic, LDS conflicts.  Counters in their own runs.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_k3_sweep
rm -rf "$OUT"; mkdir -p "$OUT"
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -o k1 -- python3 scripts/k3_sweep_once.py ${SWEEP_N:-400000} > "$OUT/$name.log" 2>&1; echo "$name rc=$?"; }
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o k1 -- python3 scripts/k3_sweep_once.py ${SWEEP_N:-400000} > "$OUT/trace.log" 2>&1
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
run rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
run lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT
run sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
python3 scripts/pmc_summary.py "$OUT" "cov_" > gpurun_out/r02_k3_sweep_rocprof_summary.txt
cat gpurun_out/r02_k3_sweep_rocprof_summary.txt
