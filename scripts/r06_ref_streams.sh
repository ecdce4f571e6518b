#!/bin/bash
# Round 6, build container only: keep the CPUs busy with whole-pipeline runs of the REFERENCE (and of this build's
# torch-module VAE on the CPU) on the two C1 stand-ins -- three streams side by side, two threads each.
#   scripts/r06_ref_streams.sh setup            side work directories (symlinks to the reads / profile files of the first)
#   scripts/r06_ref_streams.sh A|B|C            one stream (run each in the background)
# Records go to /dev/shm/streams/*.json and are folded into tests/golden/ by `make_golden_sim8.py merge`.
set -u
cd "$(dirname "$0")/.."
export OMP_NUM_THREADS=2 MKL_NUM_THREADS=2 PYTHONDONTWRITEBYTECODE=1
S=/dev/shm/streams
side() {   # side <src work> <dst work>
  [ -d "$2/out/profiles" ] && return
  mkdir -p "$2/out/profiles"
  ln -sf "$1/reads.fasta" "$2/reads.fasta"; ln -sf "$1/labels.npy" "$2/labels.npy"
  for f in "$1"/out/profiles/*; do ln -sf "$f" "$2/out/profiles/$(basename "$f")"; done
  python3 - "$2" <<'P'
import pickle, sys
w = sys.argv[1]
fa = w + "/reads.fasta"
pickle.dump({"1_1": [fa, 3], "1_2": [fa], "2_1": [fa, 32, 10], "3_1": ["numpy"]}, open(w + "/out/checkpoints", "wb"))
P
}
case "$1" in
setup)
  side /dev/shm/sim8_c1hard /dev/shm/sim8_c1hard_b; side /dev/shm/sim8_c1hard /dev/shm/sim8_c1hard_c
  side /dev/shm/sim8_c1 /dev/shm/sim8_c1_b ;;
A)
  SIM8_DATASET=c1hard SIM8_WORK=/dev/shm/sim8_c1hard SIM8_LATENTS=/dev/shm/sim8_c1hard SIM8_JSON=$S/hardA.json \
    python3 tests/golden/make_golden_sim8.py run $(seq 26 38) > $S/hardA.log 2>&1
  SIM8_DATASET=c1 SIM8_WORK=/dev/shm/sim8_c1 SIM8_LATENTS=/dev/shm/sim8_c1 SIM8_JSON=$S/c1A.json \
    python3 tests/golden/make_golden_sim8.py run $(seq 4 9) > $S/c1A.log 2>&1 ;;
B)
  SIM8_DATASET=c1hard SIM8_WORK=/dev/shm/sim8_c1hard_b SIM8_LATENTS=/dev/shm/sim8_c1hard SIM8_JSON=$S/hardB.json \
    python3 tests/golden/make_golden_sim8.py run $(seq 39 50) > $S/hardB.log 2>&1
  SIM8_DATASET=c1 SIM8_WORK=/dev/shm/sim8_c1_b SIM8_LATENTS=/dev/shm/sim8_c1 SIM8_JSON=$S/c1B.json \
    python3 tests/golden/make_golden_sim8.py run $(seq 10 15) > $S/c1B.log 2>&1 ;;
C)
  SIM8_DATASET=c1hard SIM8_WORK=/dev/shm/sim8_c1hard_c SIM8_LATENTS=/dev/shm/sim8_c1hard SIM8_JSON=$S/buildcpu_hard.json \
    python3 tests/golden/make_golden_sim8.py buildvae $(seq 1 20) > $S/buildcpu_hard.log 2>&1 ;;
esac
# (stream D, started later: the plain C1 stand-in, seeds of its own)
if [ "$1" = D ]; then
  [ -d /dev/shm/sim8_c1_d/out/profiles ] || side /dev/shm/sim8_c1 /dev/shm/sim8_c1_d
  SIM8_DATASET=c1 SIM8_WORK=/dev/shm/sim8_c1_d SIM8_LATENTS=/dev/shm/sim8_c1 SIM8_JSON=$S/c1D.json \
    python3 tests/golden/make_golden_sim8.py run $(seq 16 27) > $S/c1D.log 2>&1
fi
# (stream E, once C had finished: ten more reference runs on the hard set in C's work directory)
if [ "$1" = E ]; then
  SIM8_DATASET=c1hard SIM8_WORK=/dev/shm/sim8_c1hard_c SIM8_LATENTS=/dev/shm/sim8_c1hard SIM8_JSON=$S/hardE.json \
    python3 tests/golden/make_golden_sim8.py run $(seq 51 60) > $S/hardE.log 2>&1
fi
