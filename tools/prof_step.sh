#!/bin/bash
# bench line + rocprofv3 kernel stats of the same command:  bash tools/prof_step.sh TAG [extra bench.py args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --steps 16 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/bench_${tag}.log 2>&1
rm -rf /tmp/prof_${tag}
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${tag} -o p -- python bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-kernel-timer "$@" > gpurun_out/bench_${tag}_prof.log 2>&1
cp /tmp/prof_${tag}/p_kernel_stats.csv gpurun_out/kernel_stats_${tag}.csv 2>/dev/null || find /tmp/prof_${tag} -name "*kernel_stats.csv" -exec cp {} gpurun_out/kernel_stats_${tag}.csv \;
tail -1 gpurun_out/bench_${tag}.log | cut -c1-400
python - gpurun_out/kernel_stats_${tag}.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
iters = 21    # load phase 3 + warmup 2 + 16 timed
print('kernel time per iteration (ms): %.2f over %d launches / iteration' % (tot / 1e6 / iters, sum(int(r['Calls']) for r in rows) / iters))
for r in rows[:28]:
    print('%6.2f ms %6.1f%% %7.1f calls/it  %s' % (float(r['TotalDurationNs']) / 1e6 / iters, float(r['Percentage']), int(r['Calls']) / iters, r['Name'][:110]))
PY
