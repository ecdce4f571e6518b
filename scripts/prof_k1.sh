#!/bin/bash
# rocprofv3 passes for the K1 bench (run on the GPU box through gpurun).
# Kernel-trace/stats and each PMC group are separate runs, as the guide prescribes.
#   scripts/prof_k1.sh <tag> [extra bench args...]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${1:-r01}
shift || true
OUT=gpurun_out/prof_${TAG}
mkdir -p "$OUT"
ARGS="bench.py --steps 5 --warmup 1 --clock-ramp-ms 0 --no-cpu-baseline --no-extra --no-c4 --no-traffic $*"
# the trace pass runs the default command (clock ramp, 50 warm-up, 200 timed steps) so that its average matches bench.py's own
TRACE_ARGS="bench.py --no-cpu-baseline --no-extra --no-c4 --no-traffic $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o k1 -- python3 $TRACE_ARGS > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/pmc_lds" -o k1 -- python3 $ARGS > "$OUT/pmc_lds.log" 2>&1
echo "pmc_lds rc=$?"
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_wait" -o k1 -- python3 $ARGS > "$OUT/pmc_wait.log" 2>&1
echo "pmc_wait rc=$?"
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_THREAD_CYCLES_VALU --output-format csv -d "$OUT/pmc_mem" -o k1 -- python3 $ARGS > "$OUT/pmc_mem.log" 2>&1
echo "pmc_mem rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o k1 -- python3 $ARGS > "$OUT/pmc_fetch.log" 2>&1
echo "pmc_fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o k1 -- python3 $ARGS > "$OUT/pmc_write.log" 2>&1
echo "pmc_write rc=$?"
