cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
python3 scripts/c4_gap_probe.py 2500000
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gap_trace -o k -- python3 scripts/c4_gap_probe.py 2500000 > gpurun_out/gap.log 2>&1
tail -3 gpurun_out/gap.log
python3 scripts/kstats.py gpurun_out/gap_trace "" | head -30
