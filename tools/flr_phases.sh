#!/bin/bash
# Phase costs of the register-blocked filtered_lrelu kernel: rebuild agf_filtered_lrelu.hip with phases left out
# (-DAGF_PROFILE_PHASES=<mask>: 1 load, 2 up-FIR, 4 act, 8 down-FIR; results are wrong, only the time means something)
# and time one StyleGAN3 layer per mask.  Runs on the GPU box; restores the product build at the end.
#   tools/flr_phases.sh <out.txt> "<layer> <fwd|bwd>" ...
cd $GRAFT_REPO_ROOT
out=$1; shift
mkdir -p $(dirname $out); : > $out
for sk in 0 1 2 8 3 9 10 11 14 7; do
  touch animeface_amd/csrc/agf_filtered_lrelu.hip
  AGF_EXTRA_CXXFLAGS="-DAGF_PROFILE_PHASES=$sk" bash animeface_amd/csrc/build.sh > /dev/null 2>&1
  for cfg in "$@"; do
    echo "skip=$sk $(python tools/flr_one.py $cfg 20 16 2>&1 | tail -1)" >> $out
  done
done
touch animeface_amd/csrc/agf_filtered_lrelu.hip
bash animeface_amd/csrc/build.sh > /dev/null 2>&1
cat $out
