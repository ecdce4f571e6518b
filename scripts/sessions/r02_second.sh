#!/bin/bash
# round 2: bench under torch.distributed.run with ONE rank (RCCL initialised, the collective forced),
# both all-reduce forms, then the refreshed whole-pipeline measurements
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1
mkdir -p gpurun_out
for mode in half full; do
  LRB_ALLREDUCE=$mode timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --force-collective > gpurun_out/r02_bench_dist1_$mode.json 2> gpurun_out/r02_bench_dist1_$mode.err
  echo "dist1 $mode rc=$?"; tail -c 300 gpurun_out/r02_bench_dist1_$mode.err
done
timeout 1200 python scripts/e2e_pipeline_scale.py > gpurun_out/r02_e2e_pipeline.json 2> gpurun_out/r02_e2e_pipeline.err; echo "e2e rc=$?"
timeout 1500 python scripts/c3_full.py > gpurun_out/r02_c3_full.json 2> gpurun_out/r02_c3_full.err; echo "c3 rc=$?"
timeout 1500 python scripts/c5_full.py > gpurun_out/r02_c5_full.json 2> gpurun_out/r02_c5_full.err; echo "c5 rc=$?"
timeout 600 python scripts/bench_stages.py > gpurun_out/r02_stages.json 2> gpurun_out/r02_stages.err; echo "stages rc=$?"
