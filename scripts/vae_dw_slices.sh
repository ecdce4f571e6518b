#!/bin/bash
# step time by batch size against the number of dW slices (LRB_VAE_DW_SLICES; 0 = the rule in vae_dw_geometry)
cd "$(dirname "$0")/.."
for s in 0 4 5 7 8 12 16 32 64; do
  echo "== LRB_VAE_DW_SLICES=$s"
  if [ "$s" = 0 ]; then python3 scripts/vae_shape_probe.py; else LRB_VAE_DW_SLICES=$s python3 scripts/vae_shape_probe.py; fi
done
