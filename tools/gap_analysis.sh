#!/bin/bash
# kernel trace of graph-replayed steps: busy time vs wall span, gap statistics, per-kernel totals inside one steady-state iteration
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/gap; rocprofv3 --kernel-trace --output-format csv -d /tmp/gap -o g -- python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-r1-every-step --no-ada-variant --no-upfirdn2d-rows "$@" > /tmp/gap_bench.log 2>&1
tail -1 /tmp/gap_bench.log | cut -c1-200
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/gap/**/g_kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
# the last 8 iterations are graph replays of the GAN-loss iteration: take the window of the last ~300 ms
end = rows[-1][1]
win = [r for r in rows if r[0] > end - 300e6]
span = win[-1][1] - win[0][0]
busy = sum(e - s for s, e, _ in win)
gaps = [win[i + 1][0] - win[i][1] for i in range(len(win) - 1)]
pos = [g for g in gaps if g > 0]
print('window %.1f ms, %d kernels, busy %.1f ms (%.1f %%), positive gaps %.1f ms over %d gaps (median %.2f us, mean %.2f us)' % (
    span / 1e6, len(win), busy / 1e6, 100 * busy / span, sum(pos) / 1e6, len(pos), sorted(pos)[len(pos) // 2] / 1e3, sum(pos) / len(pos) / 1e3))
small = [(e - s) for s, e, _ in win if e - s < 10e3]
print('kernels shorter than 10 us: %d (%.1f %% of launches), their busy time %.2f ms' % (len(small), 100 * len(small) / len(win), sum(small) / 1e6))
big = sorted(gaps, reverse=True)[:5]
print('largest gaps (us):', [round(g / 1e3, 1) for g in big])
tot = collections.Counter(); cnt = collections.Counter()
for s, e, n in win:
    tot[n[:150]] += e - s; cnt[n[:150]] += 1
for n, t in tot.most_common(12):
    print('%7.2f ms %5d  %s' % (t / 1e6, cnt[n], n))
print('-- kernels with an average below 12 us, by total time (whole window)')
sm = [(t, n) for n, t in tot.items() if t / cnt[n] < 12e3]
for t, n in sorted(sm, reverse=True)[:40]:
    print('%7.2f ms %5d calls %6.1f us avg  %s' % (t / 1e6, cnt[n], t / cnt[n] / 1e3, n))
print('   total %.2f ms in %d launches' % (sum(t for t, _ in sm) / 1e6, sum(cnt[n] for _, n in sm)))
PY
