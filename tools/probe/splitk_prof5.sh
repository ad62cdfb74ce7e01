cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { label=$1; shift
  rm -rf /tmp/prof_$label
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$label -o p -- python $R/tools/time_conv.py $SHAPE > /dev/null 2>&1
  f=$(find /tmp/prof_$label -name "*kernel_stats.csv" | head -1)
  python - "$f" "$SHAPE $label" <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'conv2d_fwd' in r['Name']:
        print(sys.argv[2], 'calls', r['Calls'], 'avg_us %.2f' % (float(r['AverageNs']) / 1000), r['Name'][:75])
P
}
for SHAPE in "64 512 512 4 4" "64 512 512 8 8" "64 512 512 16 16"; do
  for d in 0 16 32 64 48 80 96 112; do run dbg$d AGF_SK=0 AGF_PF2=0 AGF_DBG=$d; done
done
