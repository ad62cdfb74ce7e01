#!/bin/bash
# the two rocprofv3 --kernel-trace --stats passes of tools/measure_r02.sh on their own (the PMC traffic pass takes ~15 minutes)
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf /tmp/prof_s; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o p -- python bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-r1-every-step --no-ada-variant --no-upfirdn2d-rows > gpurun_out/${tag}_bench_step_prof.log 2>&1
find /tmp/prof_s -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_bench_step_kernel_stats.csv \;
tail -1 gpurun_out/${tag}_bench_step_prof.log | cut -c1-200
rm -rf /tmp/prof_3; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_3 -o p -- python tools/bench_sg3.py --image-size 512 --batch 16 --steps 4 --warmup 2 > /dev/null 2>&1
find /tmp/prof_3 -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_sg3_512_kernel_stats.csv \;
