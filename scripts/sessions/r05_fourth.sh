#!/bin/bash
# round 5, fourth GPU call: the list tests (incl. the odd group cap), the multi-rank tests (rows written in place), bench
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "slice_lists or k3_sweep or from_slice or group_cap" 2>&1 | tail -8
timeout 1500 python -m pytest tests/test_gpu_multi.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -8
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_a.json 2> gpurun_out/r05_bench_a.err; echo "bench rc=$?"; tail -3 gpurun_out/r05_bench_a.err
python - <<'PY'
import json
l=json.loads(open("gpurun_out/r05_bench_a.json").read().strip().splitlines()[-1])
print("value", l["value"], "frac", l["roofline"]["frac"])
rs=l["roofline_stages"]
for k in ("k2","k3_default","k3_kept_lists"):
    print(k, {x: rs[k].get(x) for x in ("kernel_ms","frac","traffic")})
print(json.dumps(l["c4_phases"], indent=1)[:3000])
PY
