# A/B of the fragment-ahead order of conv2d_fwd_kernel (shipped) against the plain order (probe build nofa)
timeout 900 python -m pytest tests/test_hip_conv.py -x -q -m gpu 2>&1 | tail -1
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-r1-every-step --no-ada-variant --no-upfirdn2d-rows"
$B > gpurun_out/ab_fa_on.json 2> gpurun_out/ab_fa_on.err
cp animeface_amd/libagf_ops.so /tmp/keep.so; cp animeface_amd/libagf_ops_nofa.so animeface_amd/libagf_ops.so
$B > gpurun_out/ab_fa_off.json 2> gpurun_out/ab_fa_off.err
cp /tmp/keep.so animeface_amd/libagf_ops.so
$B > gpurun_out/ab_fa_on2.json 2> gpurun_out/ab_fa_on2.err
for f in on off on2; do python - gpurun_out/ab_fa_$f.json <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d['ms_per_step'], d['value'], d.get('roofline', {}).get('frac'))
P
done
