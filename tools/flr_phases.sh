#!/bin/bash
# kernel time of one filtered_lrelu layer with phases left out (AGF_FLR_SKIP) -> how the time splits.
# Needs a library built with -DAGF_PROFILE_PHASES (add it to FLAGS in animeface_amd/csrc/build.sh); the product build ignores the variable.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for sk in 0 1 2 4 8 15; do
  rm -rf /tmp/pk; AGF_FLR_SKIP=$sk rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -- python tools/flr_one.py "$@" 3 > /dev/null 2>&1
  python - $sk <<'PY'
import csv, glob, sys
f = glob.glob('/tmp/pk/*/*kernel_stats.csv')[0]
out = ['skip=%s' % sys.argv[1]]
for r in csv.DictReader(open(f)):
    if 'flr_rb' in r['Name']:
        out.append('%s avg_us=%.0f' % (r['Name'][20:52], float(r['AverageNs']) / 1e3))
print('  '.join(out))
PY
done
