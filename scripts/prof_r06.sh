#!/bin/bash
# round 6: one evidence file for every `frac` of the driver line.  rocprofv3 over bench.py's own stage set
# (`bench.py --stages-child`: K1 k = 4 / 5, K2, K3 by both routes and at 64 bins, K4, K5, the fused encoder at both shapes,
# K6's two kernels -- the calls roofline_stages times) and over the headline K1 run: kernel-trace stats, then one --pmc
# group a pass (FETCH_SIZE / WRITE_SIZE in passes of their own, as MI355X_MICROARCH.md prescribes).
# The program stands directly behind `--`.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp LRB_BENCH_CHILD=1
ST="bench.py --stages-child --reads 1000000 --read-len 10000 --no-cpu-baseline --no-extra --no-c4 --no-traffic"
K1="bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extra --no-c4 --no-traffic"
for w in stages k1; do
  OUT=gpurun_out/prof_r06_$w
  rm -rf "$OUT"; mkdir -p "$OUT"
  if [ $w = stages ]; then CMD=$ST; else CMD=$K1; fi
  run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -o k1 -- python3 $CMD > "$OUT/$name.log" 2>&1; echo "$w $name rc=$?"; }
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o k1 -- python3 $CMD > "$OUT/trace.log" 2>&1; echo "$w trace rc=$?"
  run fetch FETCH_SIZE
  run write WRITE_SIZE
  run lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT
  run sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
  run sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY
  run mfma SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA
done
# K1 at k = 4 and k = 5 WARM: the headline command with --k (clock ramp, 50 warm-up and 200 timed launches after it), so
# that the rocprof mean is the state roofline.stages.k1_k4 / k1_k5 of the line are measured in
for kk in 4 5; do
  OUT=gpurun_out/prof_r06_k1k$kk
  rm -rf "$OUT"; mkdir -p "$OUT"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o k1 -- python3 bench.py --k $kk --no-cpu-baseline --no-extra --no-c4 --no-traffic > "$OUT/trace.log" 2>&1; echo "k1k$kk trace rc=$?"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o k1 -- python3 bench.py --k $kk --steps 5 --warmup 1 --clock-ramp-ms 0 --no-cpu-baseline --no-extra --no-c4 --no-traffic > "$OUT/fetch.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o k1 -- python3 bench.py --k $kk --steps 5 --warmup 1 --clock-ramp-ms 0 --no-cpu-baseline --no-extra --no-c4 --no-traffic > "$OUT/write.log" 2>&1
  python3 scripts/pmc_summary.py $OUT "k1_" > gpurun_out/r06_k1_k${kk}_warm_rocprof_summary.txt
done
python3 scripts/pmc_summary.py gpurun_out/prof_r06_stages "seed_hist|seed_dist|gauss_assign|vae_|hdb_" > gpurun_out/r06_k4k5k6enc_rocprof_summary.txt
python3 scripts/pmc_summary.py gpurun_out/prof_r06_stages "wl_|cov_" > gpurun_out/r06_k2k3_rocprof_summary.txt
python3 scripts/pmc_summary.py gpurun_out/prof_r06_stages "k1_" > gpurun_out/r06_k1_k4k5_rocprof_summary.txt
python3 scripts/pmc_summary.py gpurun_out/prof_r06_k1 "k1_" > gpurun_out/r06_k1_lane_rocprof_summary.txt
wc -l gpurun_out/r06_*_rocprof_summary.txt
head -40 gpurun_out/r06_k4k5k6enc_rocprof_summary.txt
