#!/bin/bash
# A/B of an environment switch on the MFMA-bound conv shapes of the step:  bash tools/ab_dl.sh VAR val0 val1 ...
var=$1; shift
for shape in "128 128 128 128 128" "128 256 256 64 64" "128 512 512 32 32" "128 512 512 16 16" "64 128 128 128 128" "128 64 128 128 128" "128 128 256 64 64"; do
  for v in "$@"; do
    echo -n "$var=$v "; env $var=$v python tools/time_conv.py $shape 2>&1 | tail -1
  done
done
