cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd $R && timeout 900 python -m pytest tests/test_hip_conv.py -x -q -m gpu 2>&1 | tail -1)
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
for SHAPE in "64 512 512 4 4" "128 512 512 4 4" "64 512 512 8 8" "128 512 512 8 8" "64 512 512 16 16" "64 256 256 32 32" "64 128 128 64 64" "64 512 512 16 16 demod" "64 512 512 8 8 maskbits"; do
  run pf0 AGF_SK=0 AGF_PF2=0
  run pf2 AGF_SK=0 AGF_PF2=1
done
for SHAPE in "64 512 512 4 4" "128 512 512 4 4" "64 512 512 8 8" "128 512 512 8 8"; do
  for t in 256 512 1024; do
    run pf2_t1_T${t} AGF_SK_TILE=1 AGF_SK_T=$t AGF_SK_C=2
    run pf0_t1_T${t} AGF_SK_TILE=1 AGF_SK_T=$t AGF_SK_C=2 AGF_PF2=0
  done
  run t2_T512 AGF_SK_TILE=2 AGF_SK_T=512 AGF_SK_C=4
done
