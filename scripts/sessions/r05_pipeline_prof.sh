#!/bin/bash
# where the GPU time of a whole `lrbinner.py reads` run goes: the 432,333-read stand-in of Sim-8 with the README's flags under
# rocprofv3 --kernel-trace --stats; the kernels by total time
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out /dev/shm/pp
python3 - <<'P'
import sys; sys.path.insert(0, "tests")
from helpers import synth_sim8_c1, write_fasta
reads, labels = synth_sim8_c1()
write_fasta("/dev/shm/pp/reads.fasta", reads)
print(len(reads), "reads written")
P
rm -rf gpurun_out/pipe_trace /dev/shm/pp/out
LRB_SEED=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pipe_trace -o k -- python3 lrbinner.py reads -r /dev/shm/pp/reads.fasta -o /dev/shm/pp/out -k 3 -bc 10 -bs 32 --ae-dims 4 --ae-epochs 200 -bit 0 -mbs 5000 --cuda -t 32 > gpurun_out/pipe.log 2>&1
echo "rc=$?"; tail -2 gpurun_out/pipe.log
python3 - <<'P' | tee gpurun_out/r05_pipeline_kernel_stats.txt
import csv, glob
rows = []
for f in glob.glob("gpurun_out/pipe_trace/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# lrbinner.py reads on the 432,333-read stand-in (k=3, bc 10, bs 32, 200 epochs): {tot / 1e6:.0f} ms of kernels in all, {len(rows)} distinct kernels")
for r in rows[:28]:
    print(f'{r["Name"].split("(")[0][:60]:60s} calls={r["Calls"]:>6s} total_ms={float(r["TotalDurationNs"]) / 1e6:9.2f} avg_us={float(r["AverageNs"]) / 1e3:9.1f} {100 * float(r["TotalDurationNs"]) / tot:5.1f} %')
P
rm -rf /dev/shm/pp
