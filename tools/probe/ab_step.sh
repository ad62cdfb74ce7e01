# same-box A/B of the shipped library against a probe build on the whole step: bash tools/probe/ab_step.sh <probe name> [bench args]
name=$1; shift
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-r1-every-step --no-ada-variant --no-upfirdn2d-rows $@"
cp animeface_amd/libagf_ops.so /tmp/keep.so
for leg in new old new old; do
  if [ $leg = old ]; then cp animeface_amd/libagf_ops_$name.so animeface_amd/libagf_ops.so; else cp /tmp/keep.so animeface_amd/libagf_ops.so; fi
  $B > /tmp/ab.json 2> /tmp/ab.err
  python - $leg <<'P'
import json, sys
d = json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1])
h = d.get('roofline_conv_hbm', {})
print(sys.argv[1], d['ms_per_step'], d['value'], 'conv', d.get('roofline', {}).get('frac'), 'conv_hbm', h.get('frac'), h.get('ms_per_step'))
P
done
cp /tmp/keep.so animeface_amd/libagf_ops.so
