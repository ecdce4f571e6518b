#!/bin/bash
# round 6: after the resident-lists record got its mutex -- the CLI again and again (the crash was one run in ~40), then the suite
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out /dev/shm/vv
python3 - <<'P'
import sys; sys.path.insert(0, "tests")
from helpers import synth_sim8, write_fasta
reads, labels = synth_sim8()
write_fasta("/dev/shm/vv/reads.fasta", reads)
P
bad=0
for i in $(seq 1 60); do
  LRB_SEED=$i timeout 300 python3 lrbinner.py reads -r /dev/shm/vv/reads.fasta -o /dev/shm/vv/out -k 3 -bc 10 -bs 2 --ae-dims 4 --ae-epochs 20 -bit 0 -mbs 500 --cuda -t 32 > /dev/shm/vv/log.txt 2>&1 || { bad=$((bad+1)); echo "run $i rc=$?"; tail -3 /dev/shm/vv/log.txt | cut -c1-200; }
done
echo "60 CLI runs, $bad failed"
rm -rf /dev/shm/vv
timeout 3000 python3 -m pytest tests -q -m gpu -rf > gpurun_out/r06_gpu_suite_full.txt 2>&1; tail -6 gpurun_out/r06_gpu_suite_full.txt | tee gpurun_out/r06_gpu_suite.txt
rm -rf gpurun_out/sim8_latents
