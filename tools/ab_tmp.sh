timeout 900 python -m pytest tests/test_hip_sg2.py tests/test_hip_parity_bf16.py -x -q 2>&1 | tail -2
for i in 1 2; do python bench.py --steps 48 --no-kernel-timer --no-cpu-baseline --no-r1-every-step 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done
