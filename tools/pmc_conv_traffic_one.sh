#!/bin/bash
# HBM traffic + duration of one conv shape: bash tools/pmc_conv_traffic_one.sh N Cin Cout H W
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/pf /tmp/pw
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -o p -- python tools/time_conv.py "$@" > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -o p -- python tools/time_conv.py "$@" > /dev/null 2>&1
python - "$@" <<'PY'
import csv, sys
N, Cin, Cout, H, W = [int(v) for v in sys.argv[1:6]]
def med(path, counter):
    v = [(float(r['Counter_Value']), int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Kernel_Name'][:50]) for r in csv.DictReader(open(path)) if r['Counter_Name'] == counter and 'conv2d_fwd' in r['Kernel_Name']]
    v.sort(); return v[len(v) // 2]
f, d1, k = med('/tmp/pf/p_counter_collection.csv', 'FETCH_SIZE'); w, d2, _ = med('/tmp/pw/p_counter_collection.csv', 'WRITE_SIZE')
alg = N * H * W * (Cin + Cout) * 2
print(k, 'dur_us %.0f' % (d1 / 1e3), 'fetch_MB %.0f (x2 corr)' % (2 * f / 1024), 'write_MB %.0f' % (w / 1024), 'algorithmic_MB %.0f' % (alg / 1e6 ), 'HBM TB/s %.2f' % ((2 * f + w) * 1024 / d1 / 1e3))
PY
