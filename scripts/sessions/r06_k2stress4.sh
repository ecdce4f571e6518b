#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
run() {
  echo "== $*"
  for i in 0 1 2 3 4 5 6 7; do env "$@" STRESS_TAG=$i timeout 1200 python3 scripts/k2_stress2.py 40 20000 > gpurun_out/k2stress4_$i.log 2>&1 & done
  wait
  cat gpurun_out/k2stress4_*.log | grep 'passes:' | python3 -c "import sys,re; print('bad passes of 320:', sum(int(re.search(r'lists_same_wrong.: (\d+)', l).group(1)) for l in sys.stdin))"
}
run LRB_WL_ORDER_RUN=1
run LRB_WL_ORDER_RUN=8
run LRB_WL_ORDER_RUN=1
run LRB_WL_ORDER_RUN=8
run LRB_K3_SWEEP_READS=512
run LRB_K3_SWEEP_READS=2048
