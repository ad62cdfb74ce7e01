#!/bin/bash
# A/B of the forward-conv variants on two representative layers (HIP-event timed inside try_conv-like loop)
for v in "0 2" "0 1" "1 2" "1 1"; do
  set -- $v
  for shape in "64 512 512 32 32" "64 128 128 128 128" "64 64 64 256 256"; do
    AGF_CONV_VARIANT=$1 AGF_CONV_MT=$2 python tools/time_conv.py $shape
  done
done
