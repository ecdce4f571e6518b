#!/bin/bash
# Does the hard accuracy gate bite?  (1) as built: green.  (2) the coverage term of the VAE loss without its weight
# (ae_utils.h_params: e_cov_weight x 0 -- "a loss weight that got lost"): expected red.  (3) the coverage histogram one bin
# off for every count (cov_bin_dev: pos = c / bs instead of c / bs - 1, library rebuilt): recorded as it comes out.
# The patches are applied to the box's scratch copy of the tree only.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
T="tests/test_gpu_sim8.py::test_c1_hard_strains_vs_reference"
run() { timeout 900 python -m pytest "$T" -x -q -s 2>&1 | grep -E "C1-hard e2e|passed|failed|assert|Error" | cut -c1-220; }
echo "== (1) as built"; run | tee gpurun_out/r04_gate_demo_1.txt
echo "== (2) coverage weight of the VAE loss zeroed"
cp lrbinner_amd/ae_utils.py /tmp/ae_utils.py.orig
python3 - <<'PY'
import re
p='lrbinner_amd/ae_utils.py'
s=open(p).read()
n=s.count('w["e_cov_weight"]')
s=s.replace('w["e_cov_weight"]','(0.0 * w["e_cov_weight"])')
open(p,'w').write(s)
print("patched", n, "uses")
PY
run | tee gpurun_out/r04_gate_demo_2.txt
cp /tmp/ae_utils.py.orig lrbinner_amd/ae_utils.py
echo "== (3) coverage bin off by one"
cp lrbinner_amd/csrc/lrb_k15_dev.h /tmp/lrb_k15_dev.h.orig
sed -i 's|const uint32_t pos = c / bs - 1u;|const uint32_t pos = c / bs;|' lrbinner_amd/csrc/lrb_k15_dev.h
grep -n "const uint32_t pos = c / bs" lrbinner_amd/csrc/lrb_k15_dev.h
make -C lrbinner_amd/csrc > /tmp/make.log 2>&1 || tail -5 /tmp/make.log
run | tee gpurun_out/r04_gate_demo_3.txt
cp /tmp/lrb_k15_dev.h.orig lrbinner_amd/csrc/lrb_k15_dev.h
