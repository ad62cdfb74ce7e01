timeout 900 python -m pytest tests/test_hip_sg2.py tests/test_hip_conv.py -x -q 2>&1 | tail -3
for e in "AGF_PREP_PLAN=0" "AGF_PREP_PLAN=1" "AGF_PREP_PLAN=0" "AGF_PREP_PLAN=1"; do echo "== $e"; env $e python bench.py --steps 48 --no-kernel-timer --no-cpu-baseline --no-r1-every-step 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done
