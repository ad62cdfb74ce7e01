#!/bin/bash
# rocprofv3 kernel stats of SG3-T 512x512 B=16 iterations:  bash tools/prof_sg3.sh OUT.csv
out=${1:-gpurun_out/sg3_stats.csv}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $(dirname $out)
rm -rf /tmp/prof_3; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_3 -o p -- python tools/bench_sg3.py --image-size 512 --batch 16 --steps 4 --warmup 2 > /tmp/prof3.log 2>&1
tail -2 /tmp/prof3.log | head -1
find /tmp/prof_3 -name "*kernel_stats.csv" -exec cp {} $out \;
python3 - $out <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms / 6 iterations: %.1f' % (tot / 1e6 / 6))
for r in rows[:32]:
    print('%7.2f ms/it %5.1f%% %6.1f calls/it %8.1f us  %s' % (float(r['TotalDurationNs']) / 1e6 / 6, float(r['Percentage']), int(r['Calls']) / 6, float(r['AverageNs']) / 1e3, r['Name'][:120]))
PY
