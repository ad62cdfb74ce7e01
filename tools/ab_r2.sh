#!/bin/bash
# round-2 A/B of conv kernel switches inside the real training step (tools/conv_breakdown.py), one box.   tools/ab_r2.sh "VAR=a VAR=b" ...
mkdir -p gpurun_out
python -m pytest tests/test_hip_conv.py -x -q 2>&1 | tail -15 > gpurun_out/ab_r2_tests.txt
rm -f gpurun_out/ab_r2.txt
for cfg in "$@"; do
  echo "=== $cfg" >> gpurun_out/ab_r2.txt
  env $cfg ROWS=70 python tools/conv_breakdown.py 2>/dev/null | head -72 >> gpurun_out/ab_r2.txt
done
