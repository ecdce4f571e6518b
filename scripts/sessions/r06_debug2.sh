#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
python3 - <<'P'
import os, subprocess, sys
sys.path.insert(0, "tests")
from helpers import synth_sim8, write_fasta
reads, labels = synth_sim8()
os.makedirs("/dev/shm/dbg", exist_ok=True)
write_fasta("/dev/shm/dbg/reads.fasta", reads)
print(len(reads), "reads")
P
LRB_SEED=1 timeout 600 python3 lrbinner.py reads -r /dev/shm/dbg/reads.fasta -o /dev/shm/dbg/out -k 3 -bc 10 -bs 2 --ae-dims 4 --ae-epochs 5 -bit 0 -mbs 500 --cuda -t 32 2>&1 | tail -15 | cut -c1-400
echo "rc=$?"; tail -12 /dev/shm/dbg/out/LRBinner.log | cut -c1-400
rm -rf /dev/shm/dbg
