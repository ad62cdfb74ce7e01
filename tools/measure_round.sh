#!/bin/bash
# Tracked measurements (one box): headline bench + kernel stats, BASELINE config 2 (128x128, batch 32), StyleGAN3-T 512x512,
# kernel micro-benchmarks with the CPU rows, filtered_lrelu roofline table, conv HBM traffic inside the step.
tag=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
AGF_BENCH_SHAPES_FILE=gpurun_out/${tag}_conv_shapes.txt python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_driver_cmd.log 2>&1
tail -1 gpurun_out/${tag}_driver_cmd.log | cut -c1-300
python bench.py --steps 32 --warmup 4 --no-cpu-baseline > gpurun_out/${tag}_bench_step.log 2>&1
tail -1 gpurun_out/${tag}_bench_step.log | cut -c1-300
python bench.py --image-size 128 --batch 32 --steps 32 --warmup 4 --no-cpu-baseline > gpurun_out/${tag}_bench_128_b32.log 2>&1
tail -1 gpurun_out/${tag}_bench_128_b32.log | cut -c1-300
python tools/bench_sg3.py --image-size 512 --batch 16 --steps 16 --warmup 2 > gpurun_out/${tag}_sg3_512_b16.log 2>&1     # (16 steps: one of them carries the R1 penalty)
tail -1 gpurun_out/${tag}_sg3_512_b16.log | cut -c1-300
python tools/bench_sg3.py --image-size 256 --batch 32 --steps 16 --warmup 2 > gpurun_out/${tag}_sg3_256_b32.log 2>&1
tail -1 gpurun_out/${tag}_sg3_256_b32.log | cut -c1-200
python tools/bench_kernels.py --cpu > gpurun_out/${tag}_kernel_microbench.jsonl 2>/dev/null
python tools/bench_flrelu.py > gpurun_out/${tag}_flrelu_roofline.jsonl 2>/dev/null
cat gpurun_out/${tag}_flrelu_roofline.jsonl | cut -c1-400
HEAD=200 bash tools/iter_breakdown.sh gpurun_out/${tag}_iter_breakdown.txt > /dev/null 2>&1
rm -rf /tmp/prof_s; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o p -- python bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-r1-every-step --no-ada-variant --no-upfirdn2d-rows > gpurun_out/${tag}_bench_step_prof.log 2>&1
find /tmp/prof_s -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_bench_step_kernel_stats.csv \;
rm -rf /tmp/prof_3; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_3 -o p -- python tools/bench_sg3.py --image-size 512 --batch 16 --steps 4 --warmup 2 > /dev/null 2>&1
find /tmp/prof_3 -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_sg3_512_kernel_stats.csv \;
