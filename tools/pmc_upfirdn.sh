#!/bin/bash
# HBM traffic of the upfirdn2d / bias_act kernels from the L2 memory-side counters (FETCH_SIZE and WRITE_SIZE need separate passes;
# a third, counter-free pass gives undisturbed durations).  gfx950 correction per MI355X_MICROARCH.md: FETCH_SIZE counts 128-byte
# requests at 64 B, so wide coalesced reads are doubled.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/pf /tmp/pw /tmp/pt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -o p -- python tools/bench_kernels.py --reps 2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -o p -- python tools/bench_kernels.py --reps 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o p -- python tools/bench_kernels.py --reps 2 > /dev/null 2>&1
python - <<'PY'
import csv, collections
def load(path, counter):
    out = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter: continue
        k = r['Kernel_Name']
        if not any(s in k for s in ('upfirdn2d', 'bias_act')): continue
        out.setdefault((k, r['Grid_Size']), []).append(float(r['Counter_Value']))
    return out
f = load('/tmp/pf/p_counter_collection.csv', 'FETCH_SIZE')
w = load('/tmp/pw/p_counter_collection.csv', 'WRITE_SIZE')
dur = collections.OrderedDict()
for r in csv.DictReader(open('/tmp/pt/p_kernel_trace.csv')):
    k = r['Kernel_Name']
    if any(s in k for s in ('upfirdn2d', 'bias_act')):
        dur.setdefault((k, r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size', '')), []).append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
print('# kernel | grid | launches | fetch KB (raw) | fetch x2 (gfx950 corr.) MB | write MB | min duration us | HBM GB/s (corrected fetch + write) | frac of 8 TB/s')
durs = {}
for (k, g), v in dur.items():
    durs.setdefault(k, {})[g] = min(v)
for (k, g), fv in f.items():
    wv = w.get((k, g), [0])
    fe = sorted(fv)[len(fv)//2]; wr = sorted(wv)[len(wv)//2]
    dd = durs.get(k, {})
    d = dd.get(g) or (min(dd.values()) if dd else 0)
    tot = (2 * fe + wr) * 1024
    print('%-70s %10s n=%d fetch_raw_KB=%.0f fetch_corr_MB=%.1f write_MB=%.1f dur_us=%.1f HBM_GBps=%.0f frac=%.3f' % (k[:70], g, len(fv), fe, 2*fe/1024, wr/1024, d/1e3, tot/max(d,1), tot/max(d,1)/8000))
PY
