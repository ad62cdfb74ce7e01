#!/bin/bash
for mt in 1 2; do
  for shape in "64 512 512 32 32" "64 256 256 64 64" "64 128 128 128 128" "64 256 512 32 32" "64 512 256 32 32" "64 128 256 64 64" "64 64 128 128 128"; do
    AGF_CONV_MT=$mt python tools/time_conv.py $shape 2>/dev/null | sed "s/^/mt=$mt /" | cut -c1-120
  done
done
