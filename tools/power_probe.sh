#!/bin/bash
# sample power / clocks while a command runs: tools/power_probe.sh <cmd...>
"$@" > /tmp/pp_out.txt 2>&1 &
pid=$!
sleep ${PP_DELAY:-25}
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  rocm-smi --showpower --showclocks --showperflevel 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Socket" | tr '\n' ' ' | cut -c1-400; echo
  sleep 1
  kill -0 $pid 2>/dev/null || break
done
wait $pid
tail -1 /tmp/pp_out.txt | cut -c1-200
rocm-smi --showmaxpower 2>/dev/null | grep -i max
