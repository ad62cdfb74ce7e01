#!/bin/bash
# What a rank really gets on an 8-GPU node: 1/8 of the host cores.  One-rank RCCL group (AGF_FORCE_DP=1) pinned to 16 of the 128 cores,
# HIP-graph replay (three graphs per iteration, exchange between the launches) vs the eager loop (all-reduce from backward hooks).
out=${1:-gpurun_out/dp_host_budget.txt}; mkdir -p $(dirname $out); : > $out
nc=$(nproc); k=$((nc / 8)); [ $k -lt 1 ] && k=1
for mode in graphs eager; do
  for pin in all eighth; do
    extra=""; [ $mode = eager ] && extra="--eager"
    pre=""; [ $pin = eighth ] && pre="taskset -c 0-$((k - 1))"
    line=$(AGF_FORCE_DP=1 $pre python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-ada-variant --no-upfirdn2d-rows --no-r1-every-step $extra 2>/dev/null | tail -1)
    echo "$mode cores=$pin($([ $pin = eighth ] && echo $k || echo $nc)) $(echo "$line" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['execution'][:40], '| ms/step', d['ms_per_step'], '| p50', d['step_ms']['p50'], '| rccl', {k: (v['exposed_ms_per_step'], v['buckets'], v['bucket_mib']) for k, v in d['rccl'].items() if isinstance(v, dict)})")" >> $out
  done
done
cat $out
