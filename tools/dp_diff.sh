cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for mode in single dp; do
  if [ $mode = dp ]; then export AGF_FORCE_DP=1; fi
  rm -rf /tmp/gap_$mode; rocprofv3 --kernel-trace --output-format csv -d /tmp/gap_$mode -o g -- python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-r1-every-step --no-ada-variant --no-upfirdn2d-rows > /tmp/gap_$mode.log 2>&1
  grep '"metric"' /tmp/gap_$mode.log | cut -c100-260
done
python - <<'PY'
import csv, glob, collections
def load(mode):
    f = glob.glob('/tmp/gap_%s/**/g_kernel_trace.csv' % mode, recursive=True)[0]
    rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
    rows.sort()
    end = rows[-1][1]
    win = [r for r in rows if r[0] > end - 300e6]
    span = win[-1][1] - win[0][0]
    busy = sum(e - s for s, e, _ in win)
    tot = collections.Counter(); cnt = collections.Counter()
    for s, e, n in win:
        tot[n[:120]] += e - s; cnt[n[:120]] += 1
    gaps = sorted([win[i + 1][0] - win[i][1] for i in range(len(win) - 1)], reverse=True)
    print(mode, 'span %.1f busy %.1f kernels %d largest gaps us' % (span / 1e6, busy / 1e6, len(win)), [round(g / 1e3) for g in gaps[:8]])
    return tot, cnt
a, ac = load('single'); b, bc = load('dp')
keys = set(a) | set(b)
d = sorted(((b[k] - a[k]) / 1e6, k) for k in keys)
for x, k in d[:8] + d[-25:]:
    print('%+7.2f ms  (%d -> %d launches)  %s' % (x, ac[k], bc[k], k))
PY
