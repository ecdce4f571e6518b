#!/bin/bash
# rocprofv3 passes over K2 + K3 on the window lists (wl_part_kernel, wl_order_kernel*, wl_tally_kernel,
# wl_sweep_kernel) at 400 k x 10 kb = 4.0e9 windows: kernel times, HBM traffic (FETCH_SIZE / WRITE_SIZE in their own
# runs), L2 hits, LDS conflicts, waits, instruction mix.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_k2k3
rm -rf "$OUT"; mkdir -p "$OUT"
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -o k1 -- python3 scripts/k2k3_once.py ${K2K3_N:-400000} > "$OUT/$name.log" 2>&1; echo "$name rc=$?"; }
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o k1 -- python3 scripts/k2k3_once.py ${K2K3_N:-400000} > "$OUT/trace.log" 2>&1
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
run rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
run lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT
run sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY
python3 scripts/pmc_summary.py "$OUT" "wl_" > gpurun_out/r04_k2k3_rocprof_summary.txt
cat gpurun_out/r04_k2k3_rocprof_summary.txt
