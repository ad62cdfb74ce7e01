#!/bin/bash
for h in 0 1 0 1; do
  for shape in "64 512 512 16 16" "64 128 64 128 128" "64 64 64 128 128" "64 64 32 256 256" "64 256 256 32 32"; do
    SCALED=1 AGF_CONV_HOIST=$h python tools/time_conv.py $shape 2>/dev/null | sed "s/^/hoist=$h /" | cut -c1-130
  done
done
