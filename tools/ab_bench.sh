#!/bin/bash
# Same-box A/B of bench.py switches: alternates the variants (A B A B ...) so that clock / box drift hits both.
#   bash tools/ab_bench.sh OUT.txt "<args A>" "<args B>" [rounds]
out=$1; a=$2; b=$3; rounds=${4:-3}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $(dirname $out); : > $out
common="--steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-r1-every-step --no-ada-variant --no-upfirdn2d-rows"
for i in $(seq 1 $rounds); do
  for v in A B; do
    if [ $v = A ]; then extra="$a"; else extra="$b"; fi
    python bench.py $common $extra 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['step_ms']
print('$v [$extra] ms_per_step %.3f p50 %.3f min %.3f' % (d['ms_per_step'], s['p50'], s['min']))" | tee -a $out
  done
done
