#!/bin/bash
for ws in 0 1; do
  for shape in "64 64 64 256 256" "64 32 64 256 256" "64 64 32 256 256" "64 32 32 256 256" "64 64 64 128 128" "64 64 128 128 128"; do
    AGF_CONV_WS=$ws python tools/time_conv.py $shape 2>/dev/null | sed "s/^/ws=$ws /"
  done
done
