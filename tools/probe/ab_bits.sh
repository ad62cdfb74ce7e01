#!/bin/bash
# the fused-gradient launches with the lrelu mask as bf16 tensor vs as bits, and the producer's forward with / without the bit write
for shape in "128 64 64 256 256" "128 128 128 128 128" "128 256 256 64 64" "128 512 512 32 32" "128 512 512 16 16" "64 64 64 256 256" "64 128 128 128 128" "64 256 256 64 64"; do
  for m in mask maskbits mask maskbits; do python tools/time_conv.py $shape $m 2>&1 | tail -1; done
done
for shape in "128 32 64 256 256" "128 64 128 128 128" "128 128 256 64 64" "128 256 512 32 32"; do
  for m in bias biasbits bias biasbits; do python tools/time_conv.py $shape $m 2>&1 | tail -1; done
done
