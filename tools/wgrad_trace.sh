#!/bin/bash
# kernel-level durations of one weight-gradient shape: tools/wgrad_trace.sh "128 512 512 16 16" [ENV=..]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
shape=$1; shift
rm -rf /tmp/wt; env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wt -o t -- python tools/time_wgrad.py $shape > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/wt/**/t_kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print('%-90s calls %5s avg_us %8.1f' % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3))
PY
