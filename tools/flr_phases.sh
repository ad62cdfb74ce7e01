#!/bin/bash
# Phase costs of the register-blocked filtered_lrelu kernels: rebuild agf_filtered_lrelu.hip with phases left out
# (-DAGF_PROFILE_PHASES=<mask>: 1 load, 2 up-FIR, 4 act, 8 down-FIR; results are wrong, only the time means something)
# and report the kernels' average duration (rocprofv3 kernel trace) on one StyleGAN3 layer per mask.  Runs on the GPU box;
# restores the product build at the end.
#   [MASKS="0 11 ..."] tools/flr_phases.sh <out.txt> "<layer> <fwd|bwd>" ...   (further mask bits: 16 filter taps, 32 sign staging, 64 sum of y)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$1; shift
mkdir -p $(dirname $out); : > $out
for sk in ${MASKS:-0 11 14 13 7 1 2 8}; do
  touch animeface_amd/csrc/agf_filtered_lrelu.hip
  AGF_EXTRA_CXXFLAGS="-DAGF_PROFILE_PHASES=$sk" bash animeface_amd/csrc/build.sh > /dev/null 2>&1
  for cfg in "$@"; do
    rm -rf /tmp/fp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fp -o p -- python tools/flr_one.py $cfg 10 16 > /dev/null 2>&1
    python - "$sk" "$cfg" >> $out <<'PY'
import csv, sys
for r in csv.DictReader(open('/tmp/fp/p_kernel_stats.csv')):
    if 'flr_rb' in r['Name']:
        print('skip=%-2s %-8s %-60s avg %.1f us' % (sys.argv[1], sys.argv[2], r['Name'][:60], float(r['AverageNs']) / 1e3))
PY
  done
done
touch animeface_amd/csrc/agf_filtered_lrelu.hip
bash animeface_amd/csrc/build.sh > /dev/null 2>&1
cat $out
