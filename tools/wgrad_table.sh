#!/bin/bash
# per-layer weight-gradient times of the 256x256 step (D at the merged batch 128, G at 64 with per-image scales)
cd "$(dirname "$0")/.."
for s in "128 32 32 256 256" "128 32 64 256 256" "128 64 64 128 128" "128 64 128 128 128" "128 128 128 64 64" "128 128 256 64 64" \
         "128 256 256 32 32" "128 256 512 32 32" "128 512 512 16 16" "128 512 512 8 8" "128 512 512 4 4"; do
  env "$@" python tools/time_wgrad.py $s
done
for s in "64 64 32 256 256" "64 32 32 256 256" "64 128 64 128 128" "64 64 64 128 128" "64 256 128 64 64" "64 128 128 64 64" \
         "64 512 256 32 32" "64 256 256 32 32" "64 512 512 16 16" "64 512 512 8 8"; do
  env SCALED=1 "$@" python tools/time_wgrad.py $s
done
