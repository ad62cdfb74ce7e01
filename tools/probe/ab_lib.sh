#!/bin/bash
# A/B of probe builds (tools/probe/build_variant.sh) on the MFMA-bound conv shapes:  bash tools/probe/ab_lib.sh "" t5 t6 ...   ("" = the shipped library)
for shape in "128 128 128 128 128" "128 256 256 64 64" "128 512 512 32 32" "128 512 512 16 16" "64 128 128 128 128" "64 256 256 64 64" "128 128 256 64 64" "128 128 128 128 128 mask" "128 256 256 64 64 bias"; do
  for round in 1 2; do
  for v in "$@"; do
    echo -n "lib=${v:-shipped} "; AGF_PROBE_LIB=$v python tools/time_conv.py $shape 2>&1 | tail -1
  done; done
done
