#!/bin/bash
# Kernel trace of graph-replayed StyleGAN3-T iterations (adversarial loss only), cut at the once-per-iteration sine kernel of the Fourier-feature
# input, averaged over the last full iterations and grouped by family (cf. tools/iter_breakdown.sh for StyleGAN2).
#   bash tools/iter_breakdown_sg3.sh OUT.txt [image size] [batch] [extra bench_sg3.py arguments]
out=${1:-gpurun_out/iter_breakdown_sg3.txt}; size=${2:-512}; batch=${3:-16}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $(dirname $out)
rm -rf /tmp/itb3; rocprofv3 --kernel-trace --output-format csv -d /tmp/itb3 -o g -- python tools/bench_sg3.py --image-size $size --batch $batch --steps 12 --warmup 2 "${@:4}" > /tmp/itb3_bench.log 2>&1
tail -1 /tmp/itb3_bench.log | cut -c1-260 > $out
python - >> $out <<'PY'
import csv, glob, collections, re
f = glob.glob('/tmp/itb3/**/g_kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
marks = [i for i, r in enumerate(rows) if 'sin_kernel' in r[2]]
its = 6
a, b = marks[-its - 1], marks[-1]
win = rows[a:b]
span = (win[-1][1] - win[0][0]) / its
busy = sum(e - s for s, e, _ in win) / its
def family(n):
    if 'flr_rb' in n or 'filtered_lrelu' in n: return 'filtered_lrelu'
    if 'conv2d_wgrad' in n: return 'conv wgrad (+reduce)'
    if 'conv2d_fwd' in n or 'conv2d_pw8' in n or 'conv1x1' in n: return 'conv fwd/dgrad'
    if n.startswith('void at::') or n.startswith('at::') or 'rocclr' in n or 'elementwise' in n: return 'ATen / copies'
    if n.startswith('Cijk_') or 'rocblas' in n.lower(): return 'rocBLAS GEMM'
    if 'upfirdn' in n or 'upblur' in n: return 'FIR (upfirdn2d)'
    if 'planar_to_cl' in n or 'cl_to_planar' in n or 'cl_pad' in n: return 'layout (planar <-> channels-last, pad)'
    if 'act_bwd' in n or 'scale_dot' in n or 'bias_act' in n: return 'epilogue backward / scale_dot / bias_act'
    if 'prep_weights' in n or 'style_demod' in n or 'wsq' in n or 'ema_gain' in n or 'sum_squares' in n: return 'weight prep / style / statistics'
    if 'diffaug' in n: return 'diffaug'
    return 'other'
ft = collections.Counter(); fc = collections.Counter(); kt = collections.Counter(); kc = collections.Counter()
for s, e, n in win:
    fam = family(n); ft[fam] += e - s; fc[fam] += 1
    key = re.sub(r'\(.*', '', n)[:110]; kt[(fam, key)] += e - s; kc[(fam, key)] += 1
print('one adversarial-loss iteration (mean of %d replays): span %.2f ms, kernel busy %.2f ms (%.1f %%), %d launches' % (its, span / 1e6, busy / 1e6, 100 * busy / span, len(win) / its))
for fam, t in ft.most_common():
    print('%7.2f ms %6.1f launches  %s' % (t / 1e6 / its, fc[fam] / its, fam))
print('-- per kernel')
for (fam, key), t in kt.most_common(70):
    print('%7.3f ms %6.1f x %7.1f us  [%s] %s' % (t / 1e6 / its, kc[(fam, key)] / its, t / kc[(fam, key)] / 1e3, fam.split()[0], key))
PY
cat $out | cut -c1-200 | head -${HEAD:-40}
