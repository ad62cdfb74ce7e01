# kernel durations (rocprofv3) of the channel-sliced small-map launches: the event-timed loop of tools/time_conv.py is bound by the host's launch rate at these sizes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # label, env..., shape
  label=$1; shift
  rm -rf /tmp/prof_$label
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$label -o p -- python $R/tools/time_conv.py $SHAPE > /dev/null 2>&1
  f=$(find /tmp/prof_$label -name "*kernel_stats.csv" | head -1)
  python - "$f" "$SHAPE $label" <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'conv2d_fwd_kernel' in r['Name']:
        print(sys.argv[2], 'calls', r['Calls'], 'avg_us %.2f' % (float(r['AverageNs']) / 1000), r['Name'][:60])
P
}
if [ "${1:-}" = "depth" ]; then
  for SHAPE in "64 128 512 4 4" "64 256 512 4 4" "64 512 512 4 4" "64 1024 512 4 4" "64 128 512 8 8" "64 256 512 8 8" "64 512 512 8 8" "64 1024 512 8 8" "64 512 128 4 4" "64 512 256 4 4"; do run off AGF_SK=0; done
  exit 0
fi
for SHAPE in "64 512 512 4 4" "128 512 512 4 4" "64 512 512 8 8" "128 512 512 8 8"; do
  run off AGF_SK=0
  for tile in 0 1 2; do for t in 256 512 1024; do run t${tile}_T${t} AGF_SK_TILE=$tile AGF_SK_T=$t AGF_SK_C=2; done; done
done
