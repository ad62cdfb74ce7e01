#!/bin/bash
# Does a monitoring loop (rocm-smi polled every few seconds, as the round-end driver does: smi.*.json files) slow the timed window?
# usage: tools/bench_under_smi.sh OUTDIR
out=${1:-gpurun_out/smi}; mkdir -p $out
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/plain.log 2>&1
for mode in "-a --json" "--showuse --showpower --showclocks --json" ; do
  tag=$(echo $mode | tr -d ' -' | cut -c1-12)
  ( while true; do rocm-smi $mode > /dev/null 2>&1; sleep 1; done ) &
  pid=$!
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/smi_$tag.log 2>&1
  kill $pid; wait $pid 2>/dev/null
done
( while true; do amd-smi metric --json > /dev/null 2>&1; sleep 1; done ) &
pid=$!
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/amdsmi.log 2>&1
kill $pid; wait $pid 2>/dev/null
for f in $out/*.log; do echo $f; tail -n 1 $f | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['step_ms'])"; done
