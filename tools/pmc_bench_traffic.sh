#!/bin/bash
# HBM traffic per launch of the MFMA conv kernels inside the real training step (bench.py), from the L2 memory-side counters.
# Separate --pmc passes for FETCH_SIZE and WRITE_SIZE (they do not fit one pass); gfx950 correction: FETCH_SIZE x2 for wide
# coalesced reads (MI355X_MICROARCH.md, HBM section; validated here on bias_act whose byte count is known, tools/pmc_upfirdn.sh).
# Writes gpurun_out/conv_traffic.json; copy it to profiles/rNN_conv_fwd_traffic.json, which bench.py reads for roofline.traffic.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export AGF_BENCH_NO_LOAD_PHASE=1      # GAN-loss iterations only: the same launches as the event-timed step whose algorithmic bytes bench.py reports
rm -rf /tmp/pf /tmp/pw
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -o p -- python bench.py --steps 2 --warmup 1 --eager --no-r1-every-step --no-cpu-baseline --no-kernel-timer --no-ada-variant --no-upfirdn2d-rows > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -o p -- python bench.py --steps 2 --warmup 1 --eager --no-r1-every-step --no-cpu-baseline --no-kernel-timer --no-ada-variant --no-upfirdn2d-rows > /dev/null 2>&1
python - <<'PY'
import csv, json
def tot(path, counter, pred):
    s = n = 0
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter and pred(r['Kernel_Name']):
            s += float(r['Counter_Value']); n += 1
    return s, n
out = {}
for name, pred in [('conv2d_fwd', lambda k: 'conv2d_fwd' in k), ('conv2d_wgrad', lambda k: 'conv2d_wgrad' in k)]:
    f, nf = tot('/tmp/pf/p_counter_collection.csv', 'FETCH_SIZE', pred)
    w, nw = tot('/tmp/pw/p_counter_collection.csv', 'WRITE_SIZE', pred)
    out[name] = {'launches': nf, 'fetch_bytes_per_launch': 2 * f * 1024 / max(nf, 1), 'write_bytes_per_launch': w * 1024 / max(nw, 1),
                 'traffic_bytes_per_launch': (2 * f * 1024 / max(nf, 1)) + (w * 1024 / max(nw, 1)),
                 'note': 'FETCH_SIZE (KB) x2 gfx950 correction + WRITE_SIZE (KB), averaged over every launch of 3 GAN-loss iterations of bench.py (eager, no load phase; B=64, 256x256)'}
import os
out['commit'] = os.environ.get('AGF_COMMIT', 'unknown')      # the GPU box has no .git: pass AGF_COMMIT=$(git rev-parse --short HEAD)
json.dump(out, open('gpurun_out/conv_traffic.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
