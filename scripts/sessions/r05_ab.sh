#!/bin/bash
# same-box A/B of two builds (ab/liblrb_old.so, ab/liblrb_new.so): list tests with the new one, then kernel times in turn (PAT)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
cp ab/liblrb_new.so lrbinner_amd/liblrb_hip.so
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lists or sweep or k2 or k3 or c4 or c3" 2>&1 | grep -E "passed|failed|rror" | head -5
PAT="${PAT:-part}" bash scripts/sessions/r04_ab.sh
cp ab/liblrb_new.so lrbinner_amd/liblrb_hip.so
