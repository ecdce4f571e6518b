#!/bin/bash
# the ring form of the part kernel (LRB_WL_PART_RING=8|16) against the tile form: list tests, kernel times, counters
# (session script of commits 0b7..0065ff3, where both forms existed behind that switch; the ring form is the only one since)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for w in ${RING_W:-16}; do
  echo "== list tests, ring $w"
  LRB_WL_PART_RING=$w timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lists or sweep or k2 or k3 or c4 or c3" 2>&1 | grep -E "passed|failed|rror" | head -5
done
CFGS="${CFGS:-tile: ring8:LRB_WL_PART_RING=8 ring16:LRB_WL_PART_RING=16 tile_b: ring16_b:LRB_WL_PART_RING=16}" bash scripts/sessions/r04_time.sh 2>&1 | grep -E "rc=|part" | cut -c1-120
if [ -n "${RING_PMC:-}" ]; then
  export LRB_WL_PART_RING=$RING_PMC
  OUT=gpurun_out/prof_ring
  rm -rf "$OUT"; mkdir -p "$OUT"
  run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -o k1 -- python3 scripts/k2k3_once.py 400000 > "$OUT/$name.log" 2>&1; echo "$name rc=$?"; }
  run lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT
  run sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
  run sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY
  python3 scripts/pmc_summary.py "$OUT" "wl_part" | tee gpurun_out/r05_ring_pmc.txt
fi
