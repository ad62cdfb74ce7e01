#!/bin/bash
# Kernel trace of graph-replayed GAN-loss iterations, cut at a once-per-iteration kernel, averaged over the last full iterations and
# grouped by family: where the time of ONE iteration goes and how many launches each family costs.
#   bash tools/iter_breakdown.sh OUT.txt [extra bench.py args]
out=${1:-gpurun_out/iter_breakdown.txt}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $(dirname $out)
rm -rf /tmp/itb; rocprofv3 --kernel-trace --output-format csv -d /tmp/itb -o g -- python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-r1-every-step --no-ada-variant --no-upfirdn2d-rows "$@" > /tmp/itb_bench.log 2>&1
tail -1 /tmp/itb_bench.log | cut -c1-260 > $out
python - >> $out <<'PY'
import csv, glob, collections, re
f = glob.glob('/tmp/itb/**/g_kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
marks = [i for i, r in enumerate(rows) if 'tanh_backward' in r[2] or 'TanhBackward' in r[2]]
if len(marks) < 10:
    marks = [i for i, r in enumerate(rows) if 'diffaug' in r[2].lower()][::3]
its = 8
a, b = marks[-its - 1], marks[-1]
win = rows[a:b]
span = (win[-1][1] - win[0][0]) / its
busy = sum(e - s for s, e, _ in win) / its
def family(n):
    if 'conv2d_wgrad' in n: return 'conv wgrad (+reduce)'
    if 'conv2d_fwd' in n or 'conv2d_pw8' in n: return 'conv fwd/dgrad'
    if n.startswith('void at::') or n.startswith('at::') or 'rocclr' in n or 'elementwise' in n: return 'ATen / copies'
    if n.startswith('Cijk_') or 'rocblas' in n.lower(): return 'rocBLAS GEMM'
    if 'upfirdn' in n or 'upblur' in n: return 'FIR (upfirdn2d)'
    if 'act_bwd' in n or 'scale_dot' in n or 'bias_act' in n: return 'epilogue backward / scale_dot / bias_act'
    if 'prep_weights' in n or 'style_demod' in n or 'wsq' in n or 'modulate' in n: return 'weight prep / style'
    if 'diffaug' in n or 'planar_to_cl' in n or 'cl_to_planar' in n: return 'diffaug / layout'
    return 'other'
ft = collections.Counter(); fc = collections.Counter(); kt = collections.Counter(); kc = collections.Counter()
for s, e, n in win:
    fam = family(n); ft[fam] += e - s; fc[fam] += 1
    key = re.sub(r'\(.*', '', n)[:120]; kt[(fam, key)] += e - s; kc[(fam, key)] += 1
print('one GAN-loss iteration (mean of %d replays): span %.2f ms, kernel busy %.2f ms (%.1f %%), %d launches' % (its, span / 1e6, busy / 1e6, 100 * busy / span, len(win) / its))
for fam, t in ft.most_common():
    print('%7.2f ms %6.1f launches  %s' % (t / 1e6 / its, fc[fam] / its, fam))
print('-- per kernel')
for (fam, key), t in kt.most_common(90):
    print('%7.3f ms %6.1f x %7.1f us  [%s] %s' % (t / 1e6 / its, kc[(fam, key)] / its, t / kc[(fam, key)] / 1e3, fam.split()[0], key))
PY
cat $out | cut -c1-200 | head -${HEAD:-40}
