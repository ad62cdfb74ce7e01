#!/bin/bash
# sample power / clocks while a command runs: tools/power_probe.sh <cmd...>   (PP_DELAY seconds before the first sample, PP_N samples, PP_DT apart)
"$@" > /tmp/pp_out.txt 2>&1 &
pid=$!
sleep ${PP_DELAY:-25}
for i in $(seq 1 ${PP_N:-12}); do
  kill -0 $pid 2>/dev/null || break
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk|mclk|fclk" | sed -E 's/.*(sclk|mclk|fclk) clock level: [0-9]+: \(([0-9]+)Mhz\).*/\1 \2/; s/.*Power \(W\): ([0-9.]+).*/power \1/' | tr '\n' ' '; echo
  sleep ${PP_DT:-0.5}
done
wait $pid
tail -1 /tmp/pp_out.txt | cut -c1-200
