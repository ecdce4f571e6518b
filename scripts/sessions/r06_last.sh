#!/bin/bash
# round 6, last GPU call: the driver's order -- build check, smoke, a fast slice of the suite on the final tree
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python3 -m pytest tests -q -m gpu -x -k "not c4 and not c3_full and not c2_full and not c1_own and not c1_hard and not eight and not c5 and not one_million" 2>&1 | grep -E "passed|failed|rror" | tail -5
